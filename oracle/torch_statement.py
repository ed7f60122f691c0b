"""TEST INFRASTRUCTURE (checker and reported CPU baseline; never on the product path): the network stated a second way, from the
TRAINER's definition of it -- the TensorFlow graph that produced the weights the Rust inference path consumes -- in plain torch-CPU fp32:

    contrib/trainer/dream_tf/layers/tower.py:54-71          up-layer, blocks, heads, and the variable scopes (tensor names)
    contrib/trainer/dream_tf/layers/batch_norm.py:27-75      conv 'SAME' NHWC; folded filter dumped as transpose(HWIO, [3, 0, 1, 2]) = OHWI
    contrib/trainer/dream_tf/layers/residual_block.py:48-59  relu(alpha * (conv_2 + b_2) + (1 - alpha) * x)
    contrib/trainer/dream_tf/layers/policy_head.py:43-51     relu(conv) -> reshape(-1, 361 * S) -> matmul [361 S, 362] + offset
    contrib/trainer/dream_tf/layers/value_head.py:44-57      relu(conv, 2 planes) -> reshape(-1, 722) -> matmul [722, 1] + offset -> tanh
    contrib/trainer/dream_tf/layers/dense.py:28-39           x @ W + b with W stored [in, out]

Used by tests/test_third_statement.py (against the C oracle) and by bench.py's cpu_baseline leg as the second-opinion CPU figure
SURVEY.md 8d asks for (torch-CPU F.conv2d beside the OpenMP port).  Nothing here reads /root/reference at run time.
"""
import numpy as np
import torch
import torch.nn.functional as F

SOFTMAX_TEMPERATURE = 0.709888      # src/libdg_utils/config.rs:176-177


def trainer_graph_forward(tensors, features, batch, round_layers):
    """The trainer's tower() in inference mode on dumped (BN-folded) weights.  round_layers: round every layer output to fp16
    like the fp16 compute type does (cast_to_compute_type); False = the pure fp32 mathematics of the graph."""
    f32 = lambda name: torch.from_numpy(np.asarray(tensors[name]).astype(np.float32))
    c = int(tensors["num_channels:0"][0])
    s = int(tensors["num_samples:0"][0])
    r = (lambda t: t.half().float()) if round_layers else (lambda t: t)

    def conv(x, scope, cout, cin):
        w = f32(scope + ":0")[: cout * 9 * cin].reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)      # OHWI as dumped -> OIHW
        b = f32(scope + "/offset:0")[:cout]
        return F.conv2d(x, w, bias=None, padding=1), b.reshape(1, cout, 1, 1)                      # padding='SAME', stride 1

    x = torch.from_numpy(np.asarray(features).view(np.float16).astype(np.float32)).reshape(batch, 19, 19, 32).permute(0, 3, 1, 2)
    y, b = conv(x, "01_upsample/conv_1", c, 32)
    y = r(torch.relu(y + b))
    i = 2
    while f"{i:02d}_residual/conv_1:0" in tensors:
        scope = f"{i:02d}_residual"
        alpha = float(np.asarray(tensors[scope + "/alpha:0"]).view(np.float32)[0])
        t, b1 = conv(y, scope + "/conv_1", c, c)
        t = r(torch.relu(t + b1))
        t, b2 = conv(t, scope + "/conv_2", c, c)
        y = r(torch.relu(alpha * (t + b2) + (1.0 - alpha) * y))       # residual_block.py:56
        i += 1
    head = i
    p, pb = conv(y, f"{head:02d}p_policy/conv_1", s, c)
    p = r(torch.relu(p + pb)).permute(0, 2, 3, 1).reshape(batch, 361 * s)                           # NHWC reshape(-1, 361 * S)
    w = f32(f"{head:02d}p_policy/linear_1:0")[: 361 * s * 362].reshape(361 * s, 362)
    logits = p @ w + f32(f"{head:02d}p_policy/linear_1/offset:0")[:362]
    policy = torch.softmax(logits / SOFTMAX_TEMPERATURE, dim=1)
    v, vb = conv(y, f"{head:02d}v_value/conv_1", 2, c)
    v = r(torch.relu(v + vb)).permute(0, 2, 3, 1).reshape(batch, 722)
    w = f32(f"{head:02d}v_value/linear_2:0")[:722].reshape(722, 1)
    value = torch.tanh(v @ w + f32(f"{head:02d}v_value/linear_2/offset:0")[:1])
    return value.reshape(batch).numpy(), policy.numpy()
