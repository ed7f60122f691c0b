// Topology discovery and the packed, kernel-native weight image.
//
// Mirrors what graph::Builder::get_workspace and the layer constructors derive from the tensor map
// (src/libdg_nn/graph.rs:50-96, layers/up_block.rs:37-47, layers/residual_block.rs:40-63,
// layers/policy_head.rs:43-63, layers/value_head.rs:40-57, layers/common.rs:22-50) and performs on the
// host, once, the preparation steps the reference performs on the device at first use:
//   residual_block.rs:72-74   b2 <- fp16(alpha * b2)      (cudnnScaleTensor)
//   policy_head.rs:87-89      b_l <- fp16(tau * b_l)      (cudnnScaleTensor)
//   dense.rs:182-189          dense filter transpose      (not needed: the head kernel consumes [in][out])
#include "dgnn_internal.h"
#include "kernels.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace dgnn {

namespace {

constexpr size_t kAlign = 256;   // every tensor of the image starts on a 256-byte boundary

size_t reserve(std::vector<uint8_t>& image, size_t bytes)
{
    const size_t offset = (image.size() + kAlign - 1) / kAlign * kAlign;
    image.resize(offset + bytes, 0);
    return offset;
}

const HostTensor* find_half(const TensorMap& tensors, const std::string& name, size_t min_elements)
{
    auto it = tensors.find(name);
    if (it == tensors.end()) return nullptr;
    if (it->second.dtype != DataType::Half) return nullptr;
    if (it->second.num_elements() < min_elements) return nullptr;   // may carry base85 padding
    return &it->second;
}

// Filter [cout][3][3][cin] (KRSC, conv2d.rs:111-117) -> [9 * cin / 32][cout_pad][32]:
// k-step = tap * (cin / 32) + chunk; one k-step is one contiguous [cout_pad][32] slab.
int pack_conv(const TensorMap& tensors, const std::string& name, int cout, int cin,
              float alpha1, float alpha2, float bias_scale, unsigned layouts, std::vector<uint8_t>& image, ConvParams& out)
{
    const HostTensor* w = find_half(tensors, name + ":0", (size_t)cout * 9 * cin);
    const HostTensor* b = find_half(tensors, name + "/offset:0", (size_t)cout);
    if (!w || !b) return DGNN_MALFORMED_WEIGHTS;   // reference: expect("no filter available") panics

    const int cout_pad = (cout + 15) / 16 * 16;
    const uint16_t* src = reinterpret_cast<const uint16_t*>(w->bytes.data());

    out.cin = cin;
    out.cout = cout;
    out.cout_pad = cout_pad;
    out.alpha1 = alpha1;
    out.alpha2 = alpha2;
    // only the layouts the selected kernels read (an offset of 0 = "not packed": offset 0 is the zero page)
    out.w_offset = out.w_tile_offset = out.w_board_offset = 0;
    if (layouts & kPackGeneric) {
        out.w_offset = reserve(image, (size_t)9 * cin * cout_pad * sizeof(uint16_t));
        pack_filter(src, cout, cin, reinterpret_cast<uint16_t*>(image.data() + out.w_offset));
    }
    if ((layouts & kPackTile) && cout % 256 == 0 && cin % 64 == 0) {
        out.w_tile_offset = reserve(image, (size_t)9 * cin * cout * sizeof(uint16_t));
        pack_filter_tile(src, cout, cin, reinterpret_cast<uint16_t*>(image.data() + out.w_tile_offset));
    }
    if ((layouts & kPackBoard) && cout % 64 == 0 && cin % 32 == 0) {
        out.w_board_offset = reserve(image, board_filter_elements(cout, cin) * sizeof(uint16_t));
        pack_filter_board(src, cout, cin, reinterpret_cast<uint16_t*>(image.data() + out.w_board_offset));
    }

    out.b_offset = reserve(image, (size_t)cout_pad * sizeof(uint16_t));
    uint16_t* bias = reinterpret_cast<uint16_t*>(image.data() + out.b_offset);
    const uint16_t* bsrc = reinterpret_cast<const uint16_t*>(b->bytes.data());
    for (int co = 0; co < cout; ++co)
        bias[co] = bias_scale == 1.0f ? bsrc[co] : float_to_half(bias_scale * half_to_float(bsrc[co]));

    return DGNN_OK;
}

}  // namespace

void pack_filter(const uint16_t* krsc, int cout, int cin, uint16_t* dst)
{
    const int chunks = cin / 32;
    const int cout_pad = (cout + 15) / 16 * 16;
    for (int co = 0; co < cout; ++co)
        for (int tap = 0; tap < 9; ++tap)
            for (int ci = 0; ci < cin; ++ci) {
                const size_t kstep = (size_t)tap * chunks + ci / 32;
                dst[(kstep * cout_pad + co) * 32 + ci % 32] = krsc[((size_t)co * 9 + tap) * cin + ci];
            }
}

void pack_filter_tile(const uint16_t* krsc, int cout, int cin, uint16_t* dst)
{
    const int chunks = cin / 64;
    const int nsteps = 9 * chunks;
    for (int co = 0; co < cout; ++co) {
        const int nb = co / 256, r = co % 256;
        for (int chunk = 0; chunk < chunks; ++chunk)
            for (int tap = 0; tap < 9; ++tap) {
                const size_t step = (size_t)chunk * 9 + tap;
                uint16_t* row = dst + (((size_t)nb * nsteps + step) * 256 + r) * 64;
                const uint16_t* src = krsc + ((size_t)co * 9 + tap) * cin + chunk * 64;
                for (int s = 0; s < 8; ++s) {
                    const int piece = s ^ ((r >> 1) & 7);
                    for (int e = 0; e < 8; ++e) row[s * 8 + e] = src[piece * 8 + e];
                }
            }
    }
}

size_t board_filter_elements(int cout, int cin)
{
    return (size_t)9 * cin * cout + (size_t)9 * (cout / 32) * 1024;   // + 9 k-steps (one chunk: read ahead by the kernel, never used)
}

void pack_filter_board(const uint16_t* krsc, int cout, int cin, uint16_t* dst)
{
    const int slices = cout / 32;
    for (int chunk = 0; chunk < cin / 32; ++chunk)
        for (int tap = 0; tap < 9; ++tap)
            for (int slice = 0; slice < slices; ++slice)
                for (int half = 0; half < 2; ++half)          // the A fragment of v_mfma_f32_16x16x32_f16: 16 channels x 32 input channels,
                    for (int lane = 0; lane < 64; ++lane) {   // lane l = row l & 15, k = 8 * (l >> 4) .. + 7
                        const int co = slice * 32 + half * 16 + (lane & 15);
                        const int ci = chunk * 32 + (lane >> 4) * 8;
                        uint16_t* out = dst + (((((size_t)chunk * 9 + tap) * slices + slice) * 2 + half) * 64 + lane) * 8;
                        const uint16_t* src = krsc + ((size_t)co * 9 + tap) * cin + ci;
                        for (int e = 0; e < 8; ++e) out[e] = src[e];
                    }
}

namespace {

std::string scope(int index, const char* suffix)
{
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%02d%s", index, suffix);
    return buf;
}

}  // namespace

// Which filter layouts the image carries: by default only what the kernels of the default path read -- the board layout
// wherever the board kernel covers the network (C a multiple of 64), else the generic one.  DGNN_CONV=tile1|tile2|tile3|v1
// (read when the network is built) selects the NHWC kernels and packs their layouts instead; DGNN_PACK=all packs all three
// (A/B measurements that switch kernels per workspace).  20 x 256: 52 MB instead of 148 MB.
unsigned pack_layouts_from_env(int channels)
{
    const char* all = test_env("DGNN_PACK");
    if (all && std::strcmp(all, "all") == 0) return kPackGeneric | kPackTile | kPackBoard;
    const char* impl = std::getenv("DGNN_CONV");
    const bool board_ok = channels >= 64 && channels % 64 == 0;
    if (impl && *impl && std::strcmp(impl, "board") != 0)
        return std::strcmp(impl, "v1") == 0 ? kPackGeneric : (kPackGeneric | kPackTile);
    return board_ok ? kPackBoard : kPackGeneric;
}

namespace {

// Everything a forward pass needs besides tensors travels IN the image, so that a rank that receives it (RCCL broadcast,
// weight refresh) launches with the sender's scalars: the residual gates, the soft-max temperature and the shape the
// offsets were computed for.
struct ImageHeader {
    char magic[8];            // "DGNNIMG2"
    uint32_t header_bytes;    // sizeof(ImageHeader) + 4 * num_blocks
    uint32_t layouts;         // kPack* bits
    uint64_t image_bytes;
    int32_t num_blocks, num_channels, num_samples;
    float softmax_temperature;
    // float alpha[num_blocks] follows
};
constexpr char kMagic[8] = {'D', 'G', 'N', 'N', 'I', 'M', 'G', '2'};

}  // namespace

int adopt_image(Network& net, const uint8_t* image, size_t num_bytes)
{
    Topology& topo = net.topo;
    if (num_bytes != net.image.size() || num_bytes < kZeroPageBytes + sizeof(ImageHeader)) return DGNN_BAD_ARGUMENT;
    ImageHeader h;
    std::memcpy(&h, image + kZeroPageBytes, sizeof(h));
    if (std::memcmp(h.magic, kMagic, 8) != 0) return DGNN_MALFORMED_WEIGHTS;
    // same architecture and same packing => same offsets; anything else cannot be launched with this network's descriptors
    if (h.image_bytes != num_bytes || h.num_blocks != topo.num_blocks || h.num_channels != topo.num_channels ||
        h.num_samples != topo.num_samples || h.layouts != topo.layouts ||
        h.header_bytes != sizeof(ImageHeader) + 4u * (uint32_t)topo.num_blocks || !(h.softmax_temperature > 0.0f))
        return DGNN_BAD_ARGUMENT;
    for (int i = 0; i < topo.num_blocks; ++i) {
        float alpha;
        std::memcpy(&alpha, image + kZeroPageBytes + sizeof(ImageHeader) + 4 * (size_t)i, 4);
        topo.res2[i].alpha1 = alpha;                 // residual_block.rs:57
        topo.res2[i].alpha2 = 1.0f - alpha;
    }
    topo.softmax_temperature = h.softmax_temperature;
    topo.gates_nonzero = true;
    for (const ConvParams& p : topo.res2) topo.gates_nonzero = topo.gates_nonzero && skip_scale_foldable(p.alpha1, p.alpha2);
    if (image != net.image.data()) std::memcpy(net.image.data(), image, num_bytes);
    return DGNN_OK;
}

int build_network(Network& net)
{
    const TensorMap& t = net.tensors;
    Topology& topo = net.topo;
    std::vector<uint8_t> image;

    // layers/common.rs:26-50
    auto meta = [&](const char* name, int fallback) {
        auto it = t.find(name);
        return (it != t.end() && it->second.bytes.size() >= 4) ? (int)it->second.as_i32() : fallback;
    };
    const int c = meta("num_channels:0", 128);
    const int s = meta("num_samples:0", 8);

    if (c <= 0 || s <= 0) return DGNN_MALFORMED_WEIGHTS;
    if (c % 32 != 0 || c > 1024 || s > 16) return DGNN_UNSUPPORTED;

    reserve(image, kZeroPageBytes);   // offset 0: the zero page

    // graph.rs:76-96: probe `{:02}_residual/conv_{1,2}:0` from 2 upwards until one is missing
    int blocks = 0;
    while (t.find(scope(2 + blocks, "_residual") + "/conv_1:0") != t.end() && t.find(scope(2 + blocks, "_residual") + "/conv_2:0") != t.end()) ++blocks;
    const size_t header_offset = reserve(image, sizeof(ImageHeader) + 4 * (size_t)blocks);
    std::vector<float> alphas;

    const unsigned layouts = pack_layouts_from_env(c);
    topo.layouts = layouts;
    topo.num_channels = c;
    topo.num_samples = s;
    topo.res1.clear();
    topo.res2.clear();

    int status = pack_conv(t, "01_upsample/conv_1", c, DGNN_NUM_FEATURES, 1.0f, 0.0f, 1.0f, layouts, image, topo.up);
    if (status != DGNN_OK) return status;

    for (int i = 2; i < 2 + blocks; ++i) {
        const std::string base = scope(i, "_residual");

        float alpha = 0.5f;   // residual_block.rs:50
        auto it = t.find(base + "/alpha:0");
        if (it != t.end()) {
            if (it->second.bytes.size() < 4) return DGNN_MALFORMED_WEIGHTS;
            alpha = it->second.as_f32();
        }
        alphas.push_back(alpha);

        ConvParams c1, c2;
        status = pack_conv(t, base + "/conv_1", c, c, 1.0f, 0.0f, 1.0f, layouts, image, c1);
        if (status != DGNN_OK) return status;
        // conv_2: alpha = [gate, 1 - gate] (residual_block.rs:57) and offset scaled by gate (:72-74)
        status = pack_conv(t, base + "/conv_2", c, c, alpha, 1.0f - alpha, alpha, layouts, image, c2);
        if (status != DGNN_OK) return status;

        topo.res1.push_back(c1);
        topo.res2.push_back(c2);
    }
    topo.num_blocks = (int)topo.res1.size();
    topo.gates_nonzero = true;
    for (const ConvParams& p : topo.res2) topo.gates_nonzero = topo.gates_nonzero && skip_scale_foldable(p.alpha1, p.alpha2);

    const int head = 2 + topo.num_blocks;   // graph.rs:54-55
    const std::string policy = scope(head, "p_policy");
    const std::string value = scope(head, "v_value");

    // the head convolutions as separate generic-kernel launches: only on the NHWC paths (the board path fuses both, below)
    status = pack_conv(t, policy + "/conv_1", s, c, 1.0f, 0.0f, 1.0f, layouts & kPackGeneric, image, topo.policy_conv);
    if (status != DGNN_OK) return status;
    status = pack_conv(t, value + "/conv_1", 2, c, 1.0f, 0.0f, 1.0f, layouts & kPackGeneric, image, topo.value_conv);
    if (status != DGNN_OK) return status;

    const float tau = 1.0f / topo.softmax_temperature;   // policy_head.rs:46
    const size_t policy_in = (size_t)DGNN_NUM_POINTS * s;

    const HostTensor* pw = find_half(t, policy + "/linear_1:0", policy_in * DGNN_NUM_MOVES);
    const HostTensor* pb = find_half(t, policy + "/linear_1/offset:0", DGNN_NUM_MOVES);
    const HostTensor* vw = find_half(t, value + "/linear_2:0", (size_t)DGNN_NUM_POINTS * 2);
    const HostTensor* vb = find_half(t, value + "/linear_2/offset:0", 1);
    if (!pw || !pb || !vw || !vb) return DGNN_MALFORMED_WEIGHTS;

    topo.policy_b_offset = reserve(image, DGNN_NUM_MOVES * sizeof(uint16_t));
    {
        uint16_t* dst = reinterpret_cast<uint16_t*>(image.data() + topo.policy_b_offset);
        const uint16_t* src = reinterpret_cast<const uint16_t*>(pb->bytes.data());
        for (int i = 0; i < DGNN_NUM_MOVES; ++i) dst[i] = float_to_half(tau * half_to_float(src[i]));
    }

    topo.value_w_offset = reserve(image, (size_t)DGNN_NUM_POINTS * 2 * sizeof(uint16_t));
    std::memcpy(image.data() + topo.value_w_offset, vw->bytes.data(), (size_t)DGNN_NUM_POINTS * 2 * sizeof(uint16_t));

    topo.value_b_offset = reserve(image, sizeof(uint16_t));
    std::memcpy(image.data() + topo.value_b_offset, vb->bytes.data(), sizeof(uint16_t));

    // dense.rs:182-189: the policy matrix transposed to [out][in] for the MFMA GEMM (heads.hip), rows and columns zero padded
    topo.policy_k_pad = (int)((policy_in + 63) / 64 * 64);
    topo.policy_wt_offset = reserve(image, (size_t)384 * topo.policy_k_pad * sizeof(uint16_t));
    pack_dense_transposed(reinterpret_cast<const uint16_t*>(pw->bytes.data()), (int)policy_in, DGNN_NUM_MOVES, topo.policy_k_pad,
                          reinterpret_cast<uint16_t*>(image.data() + topo.policy_wt_offset));

    // both head convolutions as one filter of NT tiles of 16 (zero padded) output rows, in the fragment order of
    // head_conv_board_kernel: [chunk][tap][tile][lane][8], lane l = row tile * 16 + (l & 15), channels chunk * 32 + (l >> 4) * 8 ..
    topo.head_w_board_offset = 0;
    if ((layouts & kPackBoard) && c % 32 == 0 && s + 2 <= 32) {
        const HostTensor* pcw = find_half(t, policy + "/conv_1:0", (size_t)s * 9 * c);
        const HostTensor* pcb = find_half(t, policy + "/conv_1/offset:0", (size_t)s);
        const HostTensor* vcw = find_half(t, value + "/conv_1:0", (size_t)2 * 9 * c);
        const HostTensor* vcb = find_half(t, value + "/conv_1/offset:0", 2);
        if (!pcw || !pcb || !vcw || !vcb) return DGNN_MALFORMED_WEIGHTS;
        const int tiles = s + 2 <= 16 ? 1 : 2;
        topo.head_w_board_offset = reserve(image, (size_t)(c / 32) * 9 * tiles * 512 * sizeof(uint16_t));
        uint16_t* dst = reinterpret_cast<uint16_t*>(image.data() + topo.head_w_board_offset);
        const uint16_t* pwsrc = reinterpret_cast<const uint16_t*>(pcw->bytes.data());
        const uint16_t* vwsrc = reinterpret_cast<const uint16_t*>(vcw->bytes.data());
        for (int chunk = 0; chunk < c / 32; ++chunk)
            for (int tap = 0; tap < 9; ++tap)
                for (int tile = 0; tile < tiles; ++tile)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int row = tile * 16 + (lane & 15);
                        const int ci = chunk * 32 + (lane >> 4) * 8;
                        uint16_t* out = dst + ((((size_t)chunk * 9 + tap) * tiles + tile) * 64 + lane) * 8;
                        const uint16_t* src = nullptr;
                        if (row < s) src = pwsrc + ((size_t)row * 9 + tap) * c + ci;
                        else if (row < s + 2) src = vwsrc + ((size_t)(row - s) * 9 + tap) * c + ci;
                        for (int e = 0; e < 8; ++e) out[e] = src ? src[e] : 0;
                    }
        topo.head_b_offset = reserve(image, 32 * sizeof(uint16_t));
        uint16_t* hb = reinterpret_cast<uint16_t*>(image.data() + topo.head_b_offset);
        for (int i = 0; i < s; ++i) hb[i] = reinterpret_cast<const uint16_t*>(pcb->bytes.data())[i];
        for (int i = 0; i < 2; ++i) hb[s + i] = reinterpret_cast<const uint16_t*>(vcb->bytes.data())[i];
    }

    reserve(image, 0);
    image.resize((image.size() + kAlign - 1) / kAlign * kAlign, 0);

    ImageHeader h;
    std::memcpy(h.magic, kMagic, 8);
    h.header_bytes = (uint32_t)(sizeof(ImageHeader) + 4 * alphas.size());
    h.layouts = layouts;
    h.image_bytes = image.size();
    h.num_blocks = topo.num_blocks;
    h.num_channels = c;
    h.num_samples = s;
    h.softmax_temperature = topo.softmax_temperature;
    std::memcpy(image.data() + header_offset, &h, sizeof(h));
    if (!alphas.empty()) std::memcpy(image.data() + header_offset + sizeof(h), alphas.data(), 4 * alphas.size());

    net.image.swap(image);
    return DGNN_OK;
}

// dense.rs:182-189: [in][out] as stored -> [out rounded up to 32 rows][k_pad] (the MFMA GEMM's A operand), zero padded
void pack_dense_transposed(const uint16_t* w_in_out, int k_in, int n_out, int k_pad, uint16_t* dst)
{
    for (int k = 0; k < k_in; ++k)
        for (int m = 0; m < n_out; ++m) dst[(size_t)m * k_pad + k] = w_in_out[(size_t)k * n_out + m];
}

}  // namespace dgnn
