"""Python face of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``bench.py``'s ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may
import this module, and only as the checker / reported CPU baseline.  Nothing under
``dream-go_amd/`` imports it.

Contents
--------
* ``b85_decode`` / ``b85_encode_f16`` -- restatement of ``src/libdg_utils/b85.rs:101-165``.
* ``load_weights``                   -- restatement of ``src/libdg_nn/loader.rs:36-116``.
* ``OracleNetwork``                  -- name probing of ``src/libdg_nn/graph.rs:50-96`` and the layer
                                        constructors (``layers/*.rs``), bound to the C restatement
                                        ``oracle/dgnn_oracle.c`` through ctypes.
* ``forward_numpy``                  -- an independent float32 numpy/im2col statement of the same dataflow
                                        used only to cross-check the C restatement.

Whole-network parity is UNPINNED: the reference's arithmetic lives in closed-source cuDNN v8 and the
reference holds no golden policy/value vectors.  The layer-level known-answer tests of
the reference are reproduced in ``tests/test_oracle_kat.py``.
"""
import ctypes
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdgnn_oracle.so")

NUM_FEATURES = 32          # src/libdg_go/utils/features.rs:88-90
BOARD_SIZE = 19            # src/libdg_nn/layers/conv2d.rs:50
NUM_POINTS = 361
NUM_MOVES = 362            # 361 + pass, src/libdg_nn/layers/policy_head.rs:54
SOFTMAX_TEMPERATURE = 0.709888  # src/libdg_utils/config.rs:176-177
DEFAULT_NUM_CHANNELS = 128      # src/libdg_nn/layers/common.rs:22
DEFAULT_NUM_SAMPLES = 8         # src/libdg_nn/layers/common.rs:24


def build(force=False):
    """Compile oracle/dgnn_oracle.c -> oracle/libdgnn_oracle.so (gcc, see oracle/Makefile)."""
    src = os.path.join(_HERE, "dgnn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libdgnn_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


# ------------------------------------------------------------------ base85 (RFC 1924 alphabet)

_B85_ALPHABET = (b"0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"
                 b"!#$%&()*+-;<=>?@^_`{|}~")   # src/libdg_utils/b85.rs:17-30
_B85_DECODE = np.full(256, -1, dtype=np.int64)
_B85_DECODE[np.frombuffer(_B85_ALPHABET, dtype=np.uint8)] = np.arange(85)


def b85_decode(text):
    """bytes/str of base85 -> raw little-endian bytes, or None on an invalid character.

    src/libdg_utils/b85.rs:101-132: groups of five characters become one big-endian u32; a trailing
    group of fewer than five characters is silently dropped.
    """
    if isinstance(text, str):
        text = text.encode("ascii")
    digits = _B85_DECODE[np.frombuffer(text, dtype=np.uint8)]
    if (digits < 0).any():
        return None
    n_groups = len(digits) // 5
    digits = digits[:n_groups * 5].reshape(n_groups, 5)
    acc = np.zeros(n_groups, dtype=np.uint64)
    for i in range(5):
        acc = acc * np.uint64(85) + digits[:, i].astype(np.uint64)
    return (acc & np.uint64(0xFFFFFFFF)).astype(">u4").tobytes()


def b85_encode_f16(values):
    """src/libdg_utils/b85.rs:141-165 -- pairs of f32 -> f16 -> five characters."""
    raw = np.asarray(values, dtype=np.float32).astype("<f2").tobytes()
    assert len(raw) % 4 == 0
    acc = np.frombuffer(raw, dtype=">u4").astype(np.uint64)
    out = bytearray()
    for a in acc:
        a = int(a)
        out += bytes(_B85_ALPHABET[(a // 85 ** (4 - i)) % 85] for i in range(5))
    return out.decode("ascii")


_DTYPES = {"i1": np.dtype("i1"), "i4": np.dtype("<i4"), "f2": np.dtype("<f2"), "f4": np.dtype("<f4")}


class MalformedWeights(Exception):
    pass


class MissingWeights(Exception):
    pass


def load_weights(source):
    """Weights JSON -> {name: np.ndarray (1-D, as stored)} plus ``scales`` {name: float}.

    src/libdg_nn/loader.rs:36-116: top-level object; string-valued entries are ignored; every other
    entry is an object with attributes ``s`` (b85 f32 scale, parsed, never applied), ``t`` (dtype tag)
    and ``v`` (b85 payload); an empty result is MissingWeights.
    """
    if isinstance(source, (bytes, str)) and not str(source).lstrip().startswith("{"):
        if not os.path.exists(source):
            raise MissingWeights(source)
        with open(source, "r") as f:
            text = f.read()
    else:
        text = source if isinstance(source, str) else source.decode("ascii")
    if not text.strip():
        raise MissingWeights("empty")
    try:
        doc = json.loads(text)
    except json.JSONDecodeError as e:
        raise MalformedWeights(str(e))
    if not isinstance(doc, dict):
        raise MalformedWeights("top level is not an object")

    tensors, scales = {}, {}
    for name, entry in doc.items():
        if isinstance(entry, str):
            continue
        if not isinstance(entry, dict):
            raise MalformedWeights(name)
        dtype = np.dtype("<f4")   # Tensor::default() is Float (tensor.rs:50-60)
        payload = None
        for attribute, value in entry.items():
            if not isinstance(value, str):
                raise MalformedWeights(name)
            if attribute == "s":
                raw = b85_decode(value)
                if raw is None or len(raw) < 4:
                    raise MalformedWeights(name)
                scales[name] = float(np.frombuffer(raw[:4], dtype="<f4")[0])
            elif attribute == "t":
                if value not in _DTYPES:
                    raise MalformedWeights(name)
                dtype = _DTYPES[value]
            elif attribute == "v":
                payload = value
            else:
                raise MalformedWeights(name)
        # the reference decodes `v` with the dtype seen so far; writers emit s, t, v in that order
        if payload is not None:
            raw = b85_decode(payload)
            if raw is None:
                raise MalformedWeights(name)
            tensors[name] = np.frombuffer(raw, dtype=dtype).copy()
        else:
            tensors[name] = np.zeros(0, dtype=dtype)
    if not tensors:
        raise MissingWeights("no tensors")
    return tensors, scales


# ------------------------------------------------------------------ ctypes binding of the C restatement

class _CNetwork(ctypes.Structure):
    _u16p = ctypes.POINTER(ctypes.c_uint16)
    _fields_ = [
        ("num_blocks", ctypes.c_int), ("num_channels", ctypes.c_int), ("num_samples", ctypes.c_int),
        ("num_features", ctypes.c_int), ("board_size", ctypes.c_int), ("softmax_temperature", ctypes.c_float),
        ("up_w", _u16p), ("up_b", _u16p),
        ("res_w1", ctypes.POINTER(_u16p)), ("res_b1", ctypes.POINTER(_u16p)),
        ("res_w2", ctypes.POINTER(_u16p)), ("res_b2", ctypes.POINTER(_u16p)),
        ("res_alpha", ctypes.POINTER(ctypes.c_float)),
        ("policy_conv_w", _u16p), ("policy_conv_b", _u16p), ("policy_linear_w", _u16p), ("policy_linear_b", _u16p),
        ("value_conv_w", _u16p), ("value_conv_b", _u16p), ("value_linear_w", _u16p), ("value_linear_b", _u16p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        u16p = ctypes.POINTER(ctypes.c_uint16)
        L.dgo_conv3x3.argtypes = [u16p, ctypes.c_int, ctypes.c_int, ctypes.c_int, u16p, u16p, ctypes.c_int,
                                  ctypes.c_float, ctypes.c_float, u16p, ctypes.c_int, u16p]
        L.dgo_conv3x3.restype = ctypes.c_int
        L.dgo_conv3x3_plain.argtypes = L.dgo_conv3x3.argtypes
        L.dgo_conv3x3_plain.restype = ctypes.c_int
        L.dgo_dense.argtypes = [u16p, ctypes.c_int, ctypes.c_int, u16p, u16p, ctypes.c_int,
                                ctypes.c_float, ctypes.c_int, u16p]
        L.dgo_dense.restype = ctypes.c_int
        L.dgo_softmax.argtypes = [u16p, ctypes.c_int, ctypes.c_int, u16p]
        L.dgo_softmax.restype = None
        L.dgo_tanh.argtypes = [u16p, ctypes.c_size_t]
        L.dgo_tanh.restype = None
        L.dgo_scale.argtypes = [u16p, ctypes.c_size_t, ctypes.c_float]
        L.dgo_scale.restype = None
        L.dgo_transpose_in_out.argtypes = [u16p, u16p, ctypes.c_int, ctypes.c_int]
        L.dgo_transpose_in_out.restype = None
        L.dgo_forward.argtypes = [ctypes.POINTER(_CNetwork), u16p, ctypes.c_int, u16p, u16p]
        L.dgo_forward.restype = ctypes.c_int
        L.dgo_forward_ex.argtypes = [ctypes.POINTER(_CNetwork), u16p, ctypes.c_int, u16p, u16p, u16p, u16p, ctypes.c_int]
        L.dgo_forward_ex.restype = ctypes.c_int
        L.dgo_f32_to_f16.argtypes = [ctypes.c_float]
        L.dgo_f32_to_f16.restype = ctypes.c_uint16
        L.dgo_f16_to_f32.argtypes = [ctypes.c_uint16]
        L.dgo_f16_to_f32.restype = ctypes.c_float
        L.dgo_num_threads.restype = ctypes.c_int
        L.dgo_set_num_threads.argtypes = [ctypes.c_int]
        L.dgo_set_num_threads.restype = None
        # OpenMP's default is one thread per VISIBLE cpu; a GPU box of the pool shows 256 and grants 16 CPUs of time
        # (cgroup quota): 256 spinning threads on 16 CPUs turn seconds into minutes
        L.dgo_set_num_threads(effective_cpus())
        _lib = L
    return _lib


def effective_cpus():
    """The cores this process may really use: scheduler affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _u16(a):
    """Any fp16/uint16 array -> contiguous uint16 view + ctypes pointer."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.float16:
        a = a.view(np.uint16)
    assert a.dtype == np.uint16, a.dtype
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16))


def f16(values):
    """float -> fp16 bit patterns (round-to-nearest-even, as llvm.convert.to.fp16)."""
    return np.asarray(values, dtype=np.float32).astype(np.float16).view(np.uint16)


def conv3x3(x, batch, size, cin, w, bias, cout, alpha1=1.0, alpha2=0.0, z=None, relu=True, plain=False):
    """plain=True: the readable loops (dgo_conv3x3_plain) instead of the register-tiled form -- the same bits, tested."""
    x, xp = _u16(x); w, wp = _u16(w); bias, bp = _u16(bias)
    zp = None
    if z is not None:
        z, zp = _u16(z)
    y = np.empty(batch * size * size * cout, dtype=np.uint16)
    fn = lib().dgo_conv3x3_plain if plain else lib().dgo_conv3x3
    status = fn(xp, batch, size, cin, wp, bp, cout, alpha1, alpha2, zp, int(relu),
                               y.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)))
    assert status == 0, status
    return y


def dense(x, batch, num_in, w_in_out, bias, num_out, alpha1=1.0, relu=True):
    """Dense layer fed with the matrix AS STORED ([in][out]); transposes first like Dense::prepare."""
    x, xp = _u16(x); w, wp = _u16(w_in_out); bias, bp = _u16(bias)
    wt = np.empty(num_in * num_out, dtype=np.uint16)
    wtp = wt.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16))
    lib().dgo_transpose_in_out(wp, wtp, num_in, num_out)
    y = np.empty(batch * num_out, dtype=np.uint16)
    status = lib().dgo_dense(xp, batch, num_in, wtp, bp, num_out, alpha1, int(relu),
                             y.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)))
    assert status == 0, status
    return y


class OracleNetwork:
    """Weights dict -> the layer list the reference's Builder would construct (graph.rs:50-96)."""

    def __init__(self, tensors, softmax_temperature=SOFTMAX_TEMPERATURE):
        self.tensors = tensors
        self.num_channels = int(tensors["num_channels:0"][0]) if "num_channels:0" in tensors else DEFAULT_NUM_CHANNELS
        self.num_samples = int(tensors["num_samples:0"][0]) if "num_samples:0" in tensors else DEFAULT_NUM_SAMPLES
        self.softmax_temperature = float(softmax_temperature)

        def need(name):
            if name not in tensors:
                raise MissingWeights(name)   # reference: expect("no filter available") panics
            return np.ascontiguousarray(tensors[name].view(np.uint16))

        self._keep = []
        self.up_w, self.up_b = need("01_upsample/conv_1:0"), need("01_upsample/conv_1/offset:0")
        self.res = []
        i = 2
        while (f"{i:02d}_residual/conv_1:0" in tensors) and (f"{i:02d}_residual/conv_2:0" in tensors):
            alpha = tensors.get(f"{i:02d}_residual/alpha:0")
            self.res.append(dict(
                w1=need(f"{i:02d}_residual/conv_1:0"), b1=need(f"{i:02d}_residual/conv_1/offset:0"),
                w2=need(f"{i:02d}_residual/conv_2:0"), b2=need(f"{i:02d}_residual/conv_2/offset:0"),
                alpha=float(alpha.view(np.float32)[0]) if alpha is not None else 0.5))
            i += 1
        self.num_blocks = len(self.res)
        h = 2 + self.num_blocks
        self.policy_conv_w, self.policy_conv_b = need(f"{h:02d}p_policy/conv_1:0"), need(f"{h:02d}p_policy/conv_1/offset:0")
        self.policy_linear_w, self.policy_linear_b = need(f"{h:02d}p_policy/linear_1:0"), need(f"{h:02d}p_policy/linear_1/offset:0")
        self.value_conv_w, self.value_conv_b = need(f"{h:02d}v_value/conv_1:0"), need(f"{h:02d}v_value/conv_1/offset:0")
        self.value_linear_w, self.value_linear_b = need(f"{h:02d}v_value/linear_2:0"), need(f"{h:02d}v_value/linear_2/offset:0")

    def _c_struct(self):
        u16p = ctypes.POINTER(ctypes.c_uint16)
        p = lambda a: a.ctypes.data_as(u16p)
        n = max(self.num_blocks, 1)
        arr = lambda key: (u16p * n)(*[p(r[key]) for r in self.res])
        alphas = (ctypes.c_float * n)(*[r["alpha"] for r in self.res])
        c = _CNetwork(
            self.num_blocks, self.num_channels, self.num_samples, NUM_FEATURES, BOARD_SIZE, self.softmax_temperature,
            p(self.up_w), p(self.up_b), arr("w1"), arr("b1"), arr("w2"), arr("b2"), alphas,
            p(self.policy_conv_w), p(self.policy_conv_b), p(self.policy_linear_w), p(self.policy_linear_b),
            p(self.value_conv_w), p(self.value_conv_b), p(self.value_linear_w), p(self.value_linear_b))
        return c

    def forward(self, features, batch):
        """features: [batch*361*32] fp16 bits (HWC) -> (value[batch], policy[batch*362]) fp16 bits."""
        feats, fp = _u16(features)
        assert feats.size == batch * NUM_POINTS * NUM_FEATURES
        value = np.empty(batch, dtype=np.uint16)
        policy = np.empty(batch * NUM_MOVES, dtype=np.uint16)
        c = self._c_struct()
        u16p = ctypes.POINTER(ctypes.c_uint16)
        status = lib().dgo_forward(ctypes.byref(c), fp, batch, value.ctypes.data_as(u16p), policy.ctypes.data_as(u16p))
        assert status == 0, status
        return value, policy

    def forward_ex(self, features, batch, accumulate_f64=False):
        """-> (value[batch], policy[batch*362], logits[batch*362], value_pre[batch]) fp16 bits: the outputs plus the two fp16 tensors
        in front of the soft-max / tanh (policy_head.rs:95-100, value_head.rs:80-81).  accumulate_f64: exact products and sums inside
        every layer, same fp16 rounding points -- the yardstick, not the oracle."""
        feats, fp = _u16(features)
        assert feats.size == batch * NUM_POINTS * NUM_FEATURES
        value, value_pre = np.empty(batch, dtype=np.uint16), np.empty(batch, dtype=np.uint16)
        policy, logits = np.empty(batch * NUM_MOVES, dtype=np.uint16), np.empty(batch * NUM_MOVES, dtype=np.uint16)
        c = self._c_struct()
        u16p = ctypes.POINTER(ctypes.c_uint16)
        status = lib().dgo_forward_ex(ctypes.byref(c), fp, batch, value.ctypes.data_as(u16p), policy.ctypes.data_as(u16p),
                                      logits.ctypes.data_as(u16p), value_pre.ctypes.data_as(u16p), 1 if accumulate_f64 else 0)
        assert status == 0, status
        return value, policy, logits, value_pre


# ------------------------------------------------------------------ independent numpy statement (cross-check only)

def _r16(a):
    return a.astype(np.float16).astype(np.float32)


def _conv3x3_numpy(x, w_krsc, bias, alpha1=1.0, alpha2=0.0, z=None, relu=True):
    """x [B,H,W,Cin] f32, w [Cout,3,3,Cin] f32 -> fp16-rounded [B,H,W,Cout] f32."""
    b, h, wd, cin = x.shape
    cout = w_krsc.shape[0]
    xp = np.zeros((b, h + 2, wd + 2, cin), dtype=np.float32)
    xp[:, 1:-1, 1:-1, :] = x
    cols = np.concatenate([xp[:, dy:dy + h, dx:dx + wd, :] for dy in range(3) for dx in range(3)], axis=-1)
    y = cols.reshape(-1, 9 * cin) @ w_krsc.reshape(cout, 9 * cin).T
    y = np.float32(alpha1) * y + bias[None, :]
    if alpha2 != 0.0:
        y = y + np.float32(alpha2) * z.reshape(-1, cout)
    if relu:
        y = np.maximum(y, 0.0)
    return _r16(y).reshape(b, h, wd, cout)


# Winograd F(2x2, 3x3) (Lavin & Gray 2015), emulated with the operand precisions a MI355X kernel would use: the transformed
# filter U = G g G^T and the transformed input tile V = B^T d B are rounded to fp16 (MFMA operands), the 16 element-wise channel
# contractions accumulate in fp32, the output transform A^T M A is fp32.  A NUMERICS study only: is the
# 2.25x cheaper form inside the 1e-3 contract on this network?  tests/test_winograd_numerics.py, tools/winograd_numerics.py.
_WG_BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float32)
_WG_G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float32)
_WG_AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float32)


def _conv3x3_winograd(x, w_krsc, bias, alpha1=1.0, alpha2=0.0, z=None, relu=True):
    """Same contract as _conv3x3_numpy, through F(2x2, 3x3) with fp16 transformed operands and fp32 accumulation."""
    b, h, wd, cin = x.shape
    cout = w_krsc.shape[0]
    th, tw = (h + 1) // 2, (wd + 1) // 2                               # 2x2 output tiles (10 x 10 on a 19 x 19 board)
    xp = np.zeros((b, 2 * th + 2, 2 * tw + 2, cin), dtype=np.float32)
    xp[:, 1:h + 1, 1:wd + 1, :] = x
    # d[b, ty, tx, i, j, c] = xp[b, 2 ty + i, 2 tx + j, c]
    d = np.stack([np.stack([xp[:, i:i + 2 * th:2, j:j + 2 * tw:2, :] for j in range(4)], axis=3) for i in range(4)], axis=3)
    v = _r16(np.einsum("ik,btukld,jl->btuijd", _WG_BT, d, _WG_BT, optimize=True))        # B^T d B, rounded to fp16
    u = _r16(np.einsum("ik,oklc,jl->ijoc", _WG_G, w_krsc, _WG_G, optimize=True))         # G g G^T, rounded to fp16
    m = np.empty((b, th, tw, 4, 4, cout), dtype=np.float32)
    for i in range(4):
        for j in range(4):
            m[:, :, :, i, j, :] = (v[:, :, :, i, j, :].reshape(-1, cin) @ u[i, j].T).reshape(b, th, tw, cout)   # fp32 accumulate
    yt = np.einsum("ik,btuklo,jl->btuijo", _WG_AT, m, _WG_AT, optimize=True)              # A^T M A: [b, ty, tx, 2, 2, cout]
    y = yt.transpose(0, 1, 3, 2, 4, 5).reshape(b, 2 * th, 2 * tw, cout)[:, :h, :wd, :].reshape(-1, cout)
    y = np.float32(alpha1) * y + bias[None, :]
    if alpha2 != 0.0:
        y = y + np.float32(alpha2) * z.reshape(-1, cout)
    if relu:
        y = np.maximum(y, 0.0)
    return _r16(y).reshape(b, h, wd, cout)


def forward_numpy(net, features, batch, tower_conv=None):
    """Same dataflow as dgo_forward with numpy float32 GEMMs (different summation order).  tower_conv: the convolution used for
    the 2 * blocks residual layers (default the direct form; _conv3x3_winograd for the numerics study)."""
    tower = tower_conv or _conv3x3_numpy
    c, s = net.num_channels, net.num_samples
    f = lambda a: a.view(np.float16).astype(np.float32)
    x = f(np.ascontiguousarray(features).view(np.uint16)).reshape(batch, 19, 19, NUM_FEATURES)
    t = _conv3x3_numpy(x, f(net.up_w).reshape(c, 3, 3, NUM_FEATURES), f(net.up_b))
    for r in net.res:
        a = np.float32(r["alpha"])
        y = tower(t, f(r["w1"]).reshape(c, 3, 3, c), f(r["b1"]))
        b2 = _r16(a * f(r["b2"]))
        t = tower(y, f(r["w2"]).reshape(c, 3, 3, c), b2, alpha1=a, alpha2=np.float32(1.0) - a, z=t)
    v1 = _conv3x3_numpy(t, f(net.value_conv_w).reshape(2, 3, 3, c), f(net.value_conv_b)).reshape(batch, 722)
    v2 = _r16(v1 @ f(net.value_linear_w)[:722].reshape(722, 1) + f(net.value_linear_b)[:1])
    value = _r16(np.tanh(v2)).reshape(batch)
    tau = np.float32(1.0 / net.softmax_temperature)
    p1 = _conv3x3_numpy(t, f(net.policy_conv_w).reshape(s, 3, 3, c), f(net.policy_conv_b)).reshape(batch, 361 * s)
    bl = _r16(tau * f(net.policy_linear_b)[:362])
    p2 = _r16(tau * (p1 @ f(net.policy_linear_w)[:361 * s * 362].reshape(361 * s, 362)) + bl)
    e = np.exp(p2 - p2.max(axis=1, keepdims=True))
    policy = _r16(e / e.sum(axis=1, keepdims=True))
    return value.astype(np.float16).view(np.uint16), policy.reshape(-1).astype(np.float16).view(np.uint16)
