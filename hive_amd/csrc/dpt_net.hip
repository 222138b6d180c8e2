// DPT-Hybrid as ONE C-ABI object: hive_dpt_create / hive_dpt_forward / hive_dpt_destroy.
//
// Replaces `dpt.models.DPTDepthModel.forward` of the reference's (absent) third_party/dpt + timm==0.5.4 as HIVE calls it
// (/root/reference/hive/dataset_adaptors.py:1366-1374, 1419) together with the pre- and post-processing around it
// (:1407-1417 `/ 255`, NormalizeImage(0.5, 0.5); :1432-1433 uint16 millimetres; hive/io.py:1032-1039 metres, > max_depth -> 0):
// uint8 frames in HBM -> depth maps in HBM, no PyTorch in between.  Pure orchestration: every layer is one of the library's own
// kernels (csrc/stem.hip, conv.hip, dpt_ops.hip, vit.hip, dpt_head.hip); this file adds the two token-shuffling kernels and the
// activation arena.  The network's structure (module tree of isl-org/DPT `DPT` with the `vitb_rn50_384` backbone):
//   stem (7x7/2 conv, GroupNorm+ReLU, max pool) -> ResNetV2 stages 3 / 4 / 9 bottlenecks (hooks: stage 0 -> layer_1, stage 1 ->
//   layer_2) -> 1x1 patch projection + class token + position embedding -> 12 ViT blocks (hooks 8, 11) -> "project" readout ->
//   1x1 (and 3x3/2) reassemble convs -> layer{1..4}_rn -> RefineNet fusion 4..1 -> depth head -> 1 / (scale x + shift).
#include "hive_internal.hpp"

#include <algorithm>
#include <map>
#include <string>
#include <vector>

typedef unsigned short half_t;  // an element of either 16-bit type (HIVE_BF16 / HIVE_F16): the host side only sizes and offsets buffers

namespace {

// tokens[b][0] = cls + pos[0];  tokens[b][1 + i] = patch[b][i] + pos[1 + i]   (T = __bf16 or _Float16, D % 8 == 0; the adds in float, one rounding)
template <typename T>
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const T *__restrict__ patch, const T *__restrict__ cls, const T *__restrict__ pos,
                                                              T *__restrict__ tokens, int B, int n_patch, int D) {
    const int dv = D / 8;
    const long long total = (long long)B * (n_patch + 1) * dv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % dv) * 8;
        const long long row = i / dv;
        const int t = (int)(row % (n_patch + 1)), b = (int)(row / (n_patch + 1));
        const T *src = t == 0 ? cls + c : patch + ((size_t)b * n_patch + (t - 1)) * D + c;
        const uint4 ra = *reinterpret_cast<const uint4 *>(src), rp = *reinterpret_cast<const uint4 *>(pos + (size_t)t * D + c);
        const T *a = reinterpret_cast<const T *>(&ra), *p = reinterpret_cast<const T *>(&rp);
        T o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (T)((float)a[j] + (float)p[j]);
        *reinterpret_cast<uint4 *>(tokens + (size_t)row * D + c) = *reinterpret_cast<const uint4 *>(o);
    }
}

// "project" readout input: out[b][i] = concat(tokens[b][1 + i], tokens[b][0])  -> [B * n_patch][2 D]   (16-byte copies: either 16-bit type)
__global__ __launch_bounds__(256) void readout_concat_kernel(const half_t *__restrict__ tokens, half_t *__restrict__ out, int B, int n_patch, int D) {
    const int dv = D / 8;
    const long long total = (long long)B * n_patch * 2 * dv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % (2 * dv));
        const long long row = i / (2 * dv);
        const int b = (int)(row / n_patch), t = (int)(row % n_patch);
        const half_t *src = c < dv ? tokens + ((size_t)b * (n_patch + 1) + 1 + t) * D + c * 8 : tokens + (size_t)b * (n_patch + 1) * D + (c - dv) * 8;
        *reinterpret_cast<uint4 *>(out + (size_t)row * 2 * D + c * 8) = *reinterpret_cast<const uint4 *>(src);
    }
}

void launch_assemble_tokens(hive_ctx *ctx, int dtype, int blocks, const void *patch, const void *cls, const void *pos, void *tokens, int B, int n_patch, int D) {
    if (dtype == HIVE_BF16)
        hipLaunchKernelGGL(assemble_tokens_kernel<__bf16>, dim3(blocks), dim3(256), 0, ctx->stream, (const __bf16 *)patch, (const __bf16 *)cls, (const __bf16 *)pos,
                           (__bf16 *)tokens, B, n_patch, D);
    else
        hipLaunchKernelGGL(assemble_tokens_kernel<_Float16>, dim3(blocks), dim3(256), 0, ctx->stream, (const _Float16 *)patch, (const _Float16 *)cls,
                           (const _Float16 *)pos, (_Float16 *)tokens, B, n_patch, D);
}

}  // namespace

struct hive_dpt {
    hive_ctx *ctx = nullptr;
    hive_dpt_config cfg{};
    std::map<std::string, const void *> w;
    hive_vit *vit = nullptr;
    std::map<std::string, float *> gram_tables;  // per 1 x 1 convolution weight: the tables of hive_gn_gram_prepare (made on first use)
    void *arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;

    const void *get(const std::string &name) const {
        auto it = w.find(name);
        return it == w.end() ? nullptr : it->second;
    }
    // Activation arena with LIVENESS: run_forward releases a map behind its last consumer (everything runs in stream order, so a buffer
    // may be handed out again as soon as the kernel that last reads it has been LAUNCHED); alloc takes the first free block that fits,
    // else grows the high-water mark.  A dry run of the same alloc / release sequence sizes the arena (offsets are deterministic).
    // At 96 frames of 480 x 640 the network's maps total 38 GB; live at any one time: ~10 GB.
    struct Block {
        size_t off, bytes;
    };
    std::vector<Block> free_blocks;          // sorted by offset, coalesced
    std::map<const void *, Block> live;      // what alloc handed out and release has not taken back
    char *base() const { return arena ? (char *)arena : (char *)(uintptr_t)4096; }  // dry run: fake addresses, never dereferenced
    void reset_arena() {
        free_blocks.clear();
        live.clear();
        arena_used = 0;
    }
    half_t *alloc(size_t elems) {
        const size_t bytes = (elems * sizeof(half_t) + 255) & ~(size_t)255;
        size_t off = arena_used;
        bool found = false;
        for (size_t i = 0; i < free_blocks.size(); ++i)
            if (free_blocks[i].bytes >= bytes) {
                off = free_blocks[i].off;
                if (free_blocks[i].bytes == bytes)
                    free_blocks.erase(free_blocks.begin() + i);
                else
                    free_blocks[i] = Block{off + bytes, free_blocks[i].bytes - bytes};
                found = true;
                break;
            }
        if (!found) {
            if (!free_blocks.empty() && free_blocks.back().off + free_blocks.back().bytes == arena_used) {  // grow the free tail
                off = free_blocks.back().off;
                free_blocks.pop_back();
            }
            arena_used = off + bytes;
        }
        half_t *p = (half_t *)(base() + off);
        live[p] = Block{off, bytes};
        return p;
    }
    void release(const void *p) {
        if (!p) return;
        auto it = live.find(p);
        if (it == live.end()) return;  // not ours (a hook released twice, an external pointer): ignore
        Block b = it->second;
        live.erase(it);
        size_t i = 0;
        while (i < free_blocks.size() && free_blocks[i].off < b.off) ++i;
        free_blocks.insert(free_blocks.begin() + i, b);
        if (i + 1 < free_blocks.size() && free_blocks[i].off + free_blocks[i].bytes == free_blocks[i + 1].off) {
            free_blocks[i].bytes += free_blocks[i + 1].bytes;
            free_blocks.erase(free_blocks.begin() + i + 1);
        }
        if (i > 0 && free_blocks[i - 1].off + free_blocks[i - 1].bytes == free_blocks[i].off) {
            free_blocks[i - 1].bytes += free_blocks[i].bytes;
            free_blocks.erase(free_blocks.begin() + i);
        }
    }
};

namespace {

struct Map {  // a channels-last activation [N][H][W][C]
    half_t *p;
    int H, W, C;
    float *gn = nullptr;  // GroupNorm statistics of the tensor, per tile of gn_tm rows, left by the convolution that wrote it
    int gn_tm = 0;        // (hive_nhwc_conv_gn); 0: none, the GroupNorm makes its own pass
};

#define DPT_TRY(expr)              \
    do {                           \
        int _rc = (expr);          \
        if (_rc) return _rc;       \
    } while (0)

// One forward.  dry = true only walks the allocation sequence (arena sizing); the launches are skipped.
// Frames are FH x FW; the network runs at H x W (both multiples of 32).  Where the two differ the frames enter through the reference's bicubic resize
// (csrc/resize.hip) and the depth map leaves through its nearest-neighbour resize back to the frame size, hand-off included.
int run_forward(hive_dpt *d, bool dry, const uint8_t *d_rgb, int B, int FH, int FW, int H, int W, const void *d_pos, float *d_depth, float max_depth,
                uint16_t *d_mm, float *d_m) {
    hive_ctx *ctx = d->ctx;
    const bool resized = FH != H || FW != W;
    auto preprocess = [&](half_t *xin) -> int {
        if (resized) return hive_dpt_resize_preprocess(ctx, d_rgb, B, FH, FW, H, W, 0.5f, 0.5f, d->cfg.dtype, xin);
        return hive_dpt_preprocess(ctx, d_rgb, (int64_t)B * H * W * 3, 0.5f, 0.5f, d->cfg.dtype, xin);
    };
    const int dt = d->cfg.dtype;
    const std::string bb = "pretrained.model.patch_embed.backbone.";
    auto need = [&](const std::string &n, const void **out) -> int {
        *out = d->get(n);
        if (!*out) return hive_fail(ctx, HIVE_ERR_INVALID, "hive_dpt: tensor '%s' missing from the table", n.c_str());
        return HIVE_OK;
    };
    auto same_out = [](int i, int s) { return (i + s - 1) / s; };
    auto drop = [&](Map &m) {  // the map (and the GroupNorm partial sums riding on it) has no consumer left to launch
        d->release(m.p);
        d->release(m.gn);
        m.p = nullptr;
        m.gn = nullptr;
    };
    auto conv = [&](const Map &x, const std::string &wname, const char *bias_name, int cout, int k, int stride, bool same_pad, int relu, const half_t *res1,
                    const half_t *res2, bool want_relu_copy, Map *out, Map *out_relu, bool gn_stats = false) -> int {
        int pt, pl, oh, ow;
        if (same_pad) {  // timm StdConv2dSame: TensorFlow "SAME", the odd pixel at the bottom / right
            oh = same_out(x.H, stride);
            ow = same_out(x.W, stride);
            pt = std::max((oh - 1) * stride + k - x.H, 0) / 2;
            pl = std::max((ow - 1) * stride + k - x.W, 0) / 2;
        } else {  // nn.Conv2d(padding = k / 2)
            pt = pl = k / 2;
            oh = (x.H + 2 * pt - k) / stride + 1;
            ow = (x.W + 2 * pl - k) / stride + 1;
        }
        *out = Map{d->alloc((size_t)B * oh * ow * cout), oh, ow, cout};
        if (out_relu) *out_relu = Map{want_relu_copy ? d->alloc((size_t)B * oh * ow * cout) : nullptr, oh, ow, cout};
        const int64_t gn_floats = gn_stats ? hive_nhwc_conv_gn_partial_floats((int64_t)B * oh * ow, cout) : 0;
        if (gn_stats) out->gn = (float *)d->alloc((size_t)gn_floats * 2);
        if (dry) return HIVE_OK;
        const void *wp, *bp = nullptr;
        DPT_TRY(need(wname, &wp));
        if (bias_name) DPT_TRY(need(bias_name, &bp));
        if (gn_stats)  // the ResNetV2 convolutions: the GroupNorm behind each gets its statistics from this epilogue
            return hive_nhwc_conv_gn(ctx, x.p, dt, B, x.H, x.W, x.C, cout, k, stride, pt, pl, oh, ow, wp, bp, relu, res1, res2, out->p,
                                     out_relu ? out_relu->p : nullptr, out->gn, gn_floats, &out->gn_tm);
        return hive_nhwc_conv(ctx, x.p, dt, B, x.H, x.W, x.C, cout, k, stride, pt, pl, oh, ow, wp, bp, relu, res1, res2, out->p,
                              out_relu ? out_relu->p : nullptr);
    };
    // conv + GroupNorm (+ shortcut + ReLU) of a bottleneck's expanding 1 x 1 convolutions as the two-pass operation; falls back to the pair
    auto conv_norm = [&](const Map &x, const std::string &wname, const std::string &norm_prefix, int cout, int stride, const half_t *residual, int relu,
                         Map *out) -> int {
        const int oh = same_out(x.H, stride), ow = same_out(x.W, stride);
        const int64_t scratch_floats = hive_nhwc_conv_gn_partial_floats((int64_t)B * oh * ow, cout) + 2ll * B * 32;
        float *scratch = (float *)d->alloc((size_t)scratch_floats * 2);
        *out = Map{d->alloc((size_t)B * oh * ow * cout), oh, ow, cout};
        const bool eligible = cout % 256 == 0 && (cout / 32) % 8 == 0 && (long long)oh * ow >= 256;  // hive_nhwc_conv_gn_apply's conditions
        Map t{eligible ? nullptr : d->alloc((size_t)B * oh * ow * cout), oh, ow, cout};               // the pair's intermediate: small maps only
        auto done = [&](int rc) {  // scratch and the pair's intermediate are dead once the launches are queued
            d->release(scratch);
            d->release(t.p);
            return rc;
        };
        if (dry) return done(HIVE_OK);
        const void *wp, *g, *b;
        DPT_TRY(need(wname, &wp));
        DPT_TRY(need(norm_prefix + ".weight", &g));
        DPT_TRY(need(norm_prefix + ".bias", &b));
        int fused = 0;
        // GroupNorm statistics from the input's Gram matrices where that is the cheaper way to them (csrc/gram.hip; tools/probe_gram.py at the bench batch: the
        // whole operation 651 -> 532 us at 64 -> 256 channels, 378 -> 317 at 128 -> 512, 499 -> 472 for the stride-2 256 -> 512; 254 -> 262 at 256 -> 1024,
        // which keeps the two-pass form).  HIVE_GN_GRAM=0 switches it off.
        const char *gram_env = getenv("HIVE_GN_GRAM");
        // ... and only for large outputs: the three small kernels behind the Gram matrices cost 40-60 us whatever the batch (break-even at ~32 frames for the
        // 64- / 128-channel inputs, ~100 for the stride-2 one): >= 150 M output elements, 250 M for C_in = 256
        const long long out_elems = (long long)B * oh * ow * cout;
        const bool gram = !(gram_env && gram_env[0] == '0') && !ctx->deterministic && eligible &&
                          (((x.C == 64 || x.C == 128) && out_elems >= 150000000ll) || (x.C == 256 && stride == 2 && out_elems >= 250000000ll));
        if (gram) {
            float *&tables = d->gram_tables[wname];
            if (!tables) {
                HIVE_CHECK_HIP(ctx, hipMalloc((void **)&tables, (size_t)hive_gn_gram_table_floats(x.C, 32) * sizeof(float)));
                DPT_TRY(hive_gn_gram_prepare(ctx, wp, dt, x.C, cout, 32, tables));
            }
            DPT_TRY(hive_nhwc_conv_gn_apply_gram(ctx, x.p, dt, B, x.H, x.W, x.C, cout, stride, oh, ow, wp, tables, 32, g, b, d->cfg.gn_eps, residual, relu, out->p, scratch,
                                                 scratch_floats, &fused));
            if (fused) return done(HIVE_OK);
        }
        DPT_TRY(hive_nhwc_conv_gn_apply(ctx, x.p, dt, B, x.H, x.W, x.C, cout, 1, stride, 0, 0, oh, ow, wp, 32, g, b, d->cfg.gn_eps, residual, relu, out->p,
                                        scratch, scratch_floats, &fused));
        if (fused) return done(HIVE_OK);
        if (eligible) return hive_fail(ctx, HIVE_ERR_INVALID, "hive_dpt: conv + GroupNorm of '%s' was not fused", wname.c_str());
        DPT_TRY(hive_nhwc_conv_gn(ctx, x.p, dt, B, x.H, x.W, x.C, cout, 1, stride, 0, 0, oh, ow, wp, nullptr, 0, nullptr, nullptr, t.p, nullptr, scratch,
                                  scratch_floats, &t.gn_tm));
        return done(hive_nhwc_group_norm_stats(ctx, t.p, dt, B, oh * ow, cout, 32, g, b, d->cfg.gn_eps, residual, relu, out->p, scratch, t.gn_tm));
    };
    auto group_norm = [&](const Map &x, const std::string &prefix, const half_t *residual, int relu, Map *out) -> int {
        *out = Map{d->alloc((size_t)B * x.H * x.W * x.C), x.H, x.W, x.C};
        if (dry) return HIVE_OK;
        const void *g, *b;
        DPT_TRY(need(prefix + ".weight", &g));
        DPT_TRY(need(prefix + ".bias", &b));
        return hive_nhwc_group_norm_stats(ctx, x.p, dt, B, x.H * x.W, x.C, 32, g, b, d->cfg.gn_eps, residual, relu, out->p, x.gn, x.gn_tm);
    };

    Map layer_1, layer_2, layer_3, layer_4;
    auto hybrid_backbone = [&]() -> int {
        // ---- pre-processing + ResNetV2 stem -----------------------------------------------------------------------------------
        half_t *xin = d->alloc((size_t)B * H * W * 3);
        Map s0{d->alloc((size_t)B * same_out(H, 2) * same_out(W, 2) * 64), same_out(H, 2), same_out(W, 2), 64};
        const int64_t stem_floats = hive_nhwc_conv_gn_partial_floats((int64_t)B * s0.H * s0.W, 64);
        s0.gn = (float *)d->alloc((size_t)stem_floats * 2);  // the GroupNorm's sums, left by the convolution's epilogue
        if (!dry) {
            DPT_TRY(preprocess(xin));
            const void *sw;
            DPT_TRY(need(bb + "stem.conv.weight", &sw));
            DPT_TRY(hive_resnet_stem_conv_gn(ctx, xin, dt, B, H, W, sw, s0.p, s0.gn, stem_floats, &s0.gn_tm));
        }
        d->release(xin);
        // GroupNorm + ReLU + MaxPool2dSame(3, 2) in one pass: the normalised 64-channel map at half resolution never reaches memory
        Map feat{d->alloc((size_t)B * same_out(s0.H, 2) * same_out(s0.W, 2) * 64), same_out(s0.H, 2), same_out(s0.W, 2), 64};
        if (!dry) {
            const void *g, *b;
            DPT_TRY(need(bb + "stem.norm.weight", &g));
            DPT_TRY(need(bb + "stem.norm.bias", &b));
            DPT_TRY(hive_nhwc_group_norm_relu_maxpool(ctx, s0.p, dt, B, s0.H, s0.W, 64, 32, g, b, d->cfg.gn_eps, feat.p, s0.gn_tm ? s0.gn : nullptr, s0.gn_tm));
        }
        drop(s0);
        // ---- ResNetV2 stages: non pre-activation bottlenecks, GroupNorm behind every convolution ------------------------------
        const int depths[3] = {3, 4, 9}, chans[3] = {256, 512, 1024};
        Map hook[2] = {Map{nullptr, 0, 0, 0}, Map{nullptr, 0, 0, 0}};
        for (int s = 0; s < 3; ++s) {
            for (int blk = 0; blk < depths[s]; ++blk) {
                const std::string pre = bb + "stages." + std::to_string(s) + ".blocks." + std::to_string(blk) + ".";
                const int cout = chans[s], mid = cout / 4, stride = (blk == 0 && s > 0) ? 2 : 1;
                Map in = feat, shortcut = feat, t, u, t2, u2;
                if (blk == 0) DPT_TRY(conv_norm(in, pre + "downsample.conv.weight", pre + "downsample.norm", cout, stride, nullptr, 0, &shortcut));
                DPT_TRY(conv(in, pre + "conv1.weight", nullptr, mid, 1, 1, true, 0, nullptr, nullptr, false, &t, nullptr, true));
                int bneck = 0;
                // norm1 + ReLU applied while conv2 stages its input (csrc/bneck.hip): one kernel, no normalised map.  Needs conv1's sums, which its
                // epilogue leaves for maps of at least one 256-pixel tile (smaller ones take the pair below)
                if (mid == 64 && stride == 1 && (long long)t.H * t.W >= 256) {
                    t2 = Map{d->alloc((size_t)B * t.H * t.W * 64), t.H, t.W, 64};
                    const int64_t floats = hive_nhwc_conv_gn_partial_floats((int64_t)B * t.H * t.W, 64);
                    t2.gn = (float *)d->alloc((size_t)floats * 2);
                    if (dry) {
                        bneck = 1;  // (the allocation sequence must not depend on the data: the launch below always fuses for these maps)
                    } else {
                        const void *g, *b, *w2;
                        DPT_TRY(need(pre + "norm1.weight", &g));
                        DPT_TRY(need(pre + "norm1.bias", &b));
                        DPT_TRY(need(pre + "conv2.weight", &w2));
                        DPT_TRY(hive_bneck_gn_conv3x3(ctx, t.p, dt, B, t.H, t.W, 64, t.gn, t.gn_tm, g, b, d->cfg.gn_eps, w2, t2.p, t2.gn, floats, &t2.gn_tm, &bneck));
                        if (!bneck) return hive_fail(ctx, HIVE_ERR_INVALID, "hive_dpt: the 64-channel bottleneck convolution of '%s' was not fused", pre.c_str());
                    }
                    drop(t);
                }
                if (!bneck) {
                    DPT_TRY(group_norm(t, pre + "norm1", nullptr, 1, &u));
                    drop(t);
                    DPT_TRY(conv(u, pre + "conv2.weight", nullptr, mid, 3, stride, true, 0, nullptr, nullptr, false, &t2, nullptr, true));
                    drop(u);
                }
                DPT_TRY(group_norm(t2, pre + "norm2", nullptr, 1, &u2));
                drop(t2);
                DPT_TRY(conv_norm(u2, pre + "conv3.weight", pre + "norm3", cout, 1, shortcut.p, 1, &feat));  // relu(norm3(conv3(.)) + shortcut)
                drop(u2);
                if (blk == 0) drop(shortcut);  // (otherwise the shortcut IS the block's input)
                if (in.p != hook[0].p && in.p != hook[1].p) drop(in);  // the stage outputs stay: layer_1, layer_2
            }
            if (s < 2) hook[s] = feat;
        }
        layer_1 = hook[0];
        layer_2 = hook[1];

        // ---- patch projection, tokens, ViT blocks, readout ------------------------------------------------------------------------
        const int gh = feat.H, gw = feat.W, n_patch = gh * gw, N = n_patch + 1, D = 768;
        Map pe;
        DPT_TRY(conv(feat, "pretrained.model.patch_embed.proj.weight", "pretrained.model.patch_embed.proj.bias", D, 1, 1, false, 0, nullptr, nullptr, false, &pe,
                     nullptr));
        half_t *tokens = d->alloc((size_t)B * N * D), *tap3 = d->alloc((size_t)B * N * D), *tap4 = d->alloc((size_t)B * N * D);
        half_t *cat = d->alloc((size_t)B * n_patch * 2 * D);
        Map map3{d->alloc((size_t)B * n_patch * D), gh, gw, D}, map4{d->alloc((size_t)B * n_patch * D), gh, gw, D};
        if (!dry) {
            const void *cls;
            DPT_TRY(need("pretrained.model.cls_token", &cls));
            const int blocks = (int)std::min<long long>(((long long)B * N * D / 8 + 255) / 256, (long long)ctx->num_cus * 16);
            launch_assemble_tokens(ctx, dt, blocks, pe.p, cls, d_pos, tokens, B, n_patch, D);
            HIVE_CHECK_HIP(ctx, hipGetLastError());
            const int taps[2] = {8, 11};
            void *tap_out[2] = {tap3, tap4};
            DPT_TRY(hive_vit_forward(d->vit, tokens, B, N, taps, 2, tap_out));
            const half_t *tap[2] = {tap3, tap4};
            Map *maps[2] = {&map3, &map4};
            for (int r = 0; r < 2; ++r) {
                hipLaunchKernelGGL(readout_concat_kernel, dim3(blocks), dim3(256), 0, ctx->stream, tap[r], cat, B, n_patch, D);
                HIVE_CHECK_HIP(ctx, hipGetLastError());
                const std::string pre = std::string("pretrained.act_postprocess") + (r == 0 ? "3" : "4") + ".0.project.0.";
                const void *rw, *rb;
                DPT_TRY(need(pre + "weight", &rw));
                DPT_TRY(need(pre + "bias", &rb));  // float32
                DPT_TRY(hive_vit_linear(ctx, cat, dt, rw, (const float *)rb, nullptr, maps[r]->p, B * n_patch, D, 2 * D, 1 /* GELU */));
            }
        }
        drop(feat);  // the last stage's output fed the patch projection only
        drop(pe);
        d->release(tokens);
        d->release(tap3);
        d->release(tap4);
        d->release(cat);
        Map t4;
        DPT_TRY(conv(map3, "pretrained.act_postprocess3.3.weight", "pretrained.act_postprocess3.3.bias", D, 1, 1, false, 0, nullptr, nullptr, false, &layer_3, nullptr));
        drop(map3);
        DPT_TRY(conv(map4, "pretrained.act_postprocess4.3.weight", "pretrained.act_postprocess4.3.bias", D, 1, 1, false, 0, nullptr, nullptr, false, &t4, nullptr));
        drop(map4);
        DPT_TRY(conv(t4, "pretrained.act_postprocess4.4.weight", "pretrained.act_postprocess4.4.bias", D, 3, 2, false, 0, nullptr, nullptr, false, &layer_4, nullptr));
        drop(t4);
        return HIVE_OK;
    };
    // ---- DPT-Large backbone (timm vit_large_patch16_384 as isl-org/DPT hooks it: blocks 5 / 11 / 17 / 23) -------------------------
    // pre-processing -> 16 x 16 / 16 patch embedding (rows + GEMM) -> class token + position embedding -> 24 ViT blocks -> "project"
    // readouts -> reassemble: 1x1 (+ ConvTranspose 4x4/4 | ConvTranspose 2x2/2 | nothing | 3x3/2) to 256 / 512 / 1024 / 1024 channels
    auto large_backbone = [&]() -> int {
        const int D = 1024, gh = H / 16, gw = W / 16, n_patch = gh * gw, N = n_patch + 1;
        half_t *xin = d->alloc((size_t)B * H * W * 3), *cols = d->alloc((size_t)B * n_patch * 768), *pe = d->alloc((size_t)B * n_patch * D);
        half_t *tokens = d->alloc((size_t)B * N * D), *cat = d->alloc((size_t)B * n_patch * 2 * D);
        half_t *tapb[4];
        Map maps[4];
        for (int r = 0; r < 4; ++r) {
            tapb[r] = d->alloc((size_t)B * N * D);
            maps[r] = Map{d->alloc((size_t)B * n_patch * D), gh, gw, D};
        }
        if (!dry) {
            DPT_TRY(preprocess(xin));
            DPT_TRY(hive_patch_rows(ctx, xin, dt, B, H, W, 3, 16, cols));
            const void *pw, *pb, *cls;
            DPT_TRY(need("pretrained.model.patch_embed.proj.weight", &pw));    // [D][16][16][3] = [D][768]
            DPT_TRY(need("pretrained.model.patch_embed.proj.bias.f32", &pb));
            DPT_TRY(need("pretrained.model.cls_token", &cls));
            DPT_TRY(hive_vit_linear(ctx, cols, dt, pw, (const float *)pb, nullptr, pe, B * n_patch, D, 768, 0));
            const int blocks = (int)std::min<long long>(((long long)B * N * D / 8 + 255) / 256, (long long)ctx->num_cus * 16);
            launch_assemble_tokens(ctx, dt, blocks, pe, cls, d_pos, tokens, B, n_patch, D);
            HIVE_CHECK_HIP(ctx, hipGetLastError());
            const int taps[4] = {5, 11, 17, 23};
            void *tap_out[4] = {tapb[0], tapb[1], tapb[2], tapb[3]};
            DPT_TRY(hive_vit_forward(d->vit, tokens, B, N, taps, 4, tap_out));
            for (int r = 0; r < 4; ++r) {
                hipLaunchKernelGGL(readout_concat_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const half_t *)tapb[r], cat, B, n_patch, D);
                HIVE_CHECK_HIP(ctx, hipGetLastError());
                const std::string pre = "pretrained.act_postprocess" + std::to_string(r + 1) + ".0.project.0.";
                const void *rw, *rb;
                DPT_TRY(need(pre + "weight", &rw));
                DPT_TRY(need(pre + "bias", &rb));  // float32
                DPT_TRY(hive_vit_linear(ctx, cat, dt, rw, (const float *)rb, nullptr, maps[r].p, B * n_patch, D, 2 * D, 1 /* GELU */));
            }
        }
        d->release(xin);
        d->release(cols);
        d->release(pe);
        d->release(tokens);
        d->release(cat);
        for (int r = 0; r < 4; ++r) d->release(tapb[r]);
        // ConvTranspose2d(C, C, s, s) = 1 x 1 convolution to s s C channels ((dy, dx, co) rows of the re-laid-out weight) + scatter + bias
        auto conv_transpose = [&](const Map &x, const std::string &prefix, int s_, Map *out) -> int {
            Map tmp{d->alloc((size_t)B * x.H * x.W * s_ * s_ * x.C), x.H, x.W, s_ * s_ * x.C};
            *out = Map{d->alloc((size_t)B * x.H * x.W * s_ * s_ * x.C), s_ * x.H, s_ * x.W, x.C};
            if (dry) {
                drop(tmp);
                return HIVE_OK;
            }
            const void *wr, *bias;
            DPT_TRY(need(prefix + ".weight.rows", &wr));
            DPT_TRY(need(prefix + ".bias", &bias));
            DPT_TRY(hive_nhwc_conv(ctx, x.p, dt, B, x.H, x.W, x.C, s_ * s_ * x.C, 1, 1, 0, 0, x.H, x.W, wr, nullptr, 0, nullptr, nullptr, tmp.p, nullptr));
            const int rc = hive_nhwc_pixel_shuffle_bias(ctx, tmp.p, bias, dt, B, x.H, x.W, x.C, s_, out->p);
            drop(tmp);
            return rc;
        };
        const std::string pp = "pretrained.act_postprocess";
        Map t;
        DPT_TRY(conv(maps[0], pp + "1.3.weight", (pp + "1.3.bias").c_str(), 256, 1, 1, false, 0, nullptr, nullptr, false, &t, nullptr));
        drop(maps[0]);
        DPT_TRY(conv_transpose(t, pp + "1.4", 4, &layer_1));
        drop(t);
        DPT_TRY(conv(maps[1], pp + "2.3.weight", (pp + "2.3.bias").c_str(), 512, 1, 1, false, 0, nullptr, nullptr, false, &t, nullptr));
        drop(maps[1]);
        DPT_TRY(conv_transpose(t, pp + "2.4", 2, &layer_2));
        drop(t);
        DPT_TRY(conv(maps[2], pp + "3.3.weight", (pp + "3.3.bias").c_str(), 1024, 1, 1, false, 0, nullptr, nullptr, false, &layer_3, nullptr));
        drop(maps[2]);
        DPT_TRY(conv(maps[3], pp + "4.3.weight", (pp + "4.3.bias").c_str(), 1024, 1, 1, false, 0, nullptr, nullptr, false, &t, nullptr));
        drop(maps[3]);
        DPT_TRY(conv(t, pp + "4.4.weight", (pp + "4.4.bias").c_str(), 1024, 3, 2, false, 0, nullptr, nullptr, false, &layer_4, nullptr));
        drop(t);
        return HIVE_OK;
    };

    DPT_TRY(d->cfg.backbone == 0 ? hybrid_backbone() : large_backbone());

    // ---- decoder: layerN_rn, four RefineNet fusion blocks --------------------------------------------------------------------
    auto refinenet = [&](int n, const Map *path, const Map &lrn, const Map &lrn_relu, Map *out) -> int {
        const std::string pre = "scratch.refinenet" + std::to_string(n) + ".";
        Map o = lrn, o_relu = lrn_relu, t;
        if (path) {  // output = path + resConfUnit1(layer_rn): conv2(relu(conv1(relu(x)))) + x + path, and its ReLU for the next unit
            DPT_TRY(conv(lrn_relu, pre + "resConfUnit1.conv1.weight", (pre + "resConfUnit1.conv1.bias").c_str(), 256, 3, 1, false, 1, nullptr, nullptr, false, &t,
                         nullptr));
            DPT_TRY(conv(t, pre + "resConfUnit1.conv2.weight", (pre + "resConfUnit1.conv2.bias").c_str(), 256, 3, 1, false, 0, lrn.p, path->p, true, &o, &o_relu));
            drop(t);
        }
        Map u, low, t2;
        DPT_TRY(conv(o_relu, pre + "resConfUnit2.conv1.weight", (pre + "resConfUnit2.conv1.bias").c_str(), 256, 3, 1, false, 1, nullptr, nullptr, false, &t2, nullptr));
        DPT_TRY(conv(t2, pre + "resConfUnit2.conv2.weight", (pre + "resConfUnit2.conv2.bias").c_str(), 256, 3, 1, false, 0, o.p, nullptr, false, &u, nullptr));
        drop(t2);
        if (o.p != lrn.p) drop(o);  // (with a path, o / o_relu are resConfUnit1's outputs; without, they ARE lrn / lrn_relu, dropped by the caller)
        if (o_relu.p != lrn_relu.p) drop(o_relu);
        // the 1x1 out_conv commutes with the bilinear interpolation: it runs on a quarter of the pixels, its bias is added on load
        DPT_TRY(conv(u, pre + "out_conv.weight", nullptr, 256, 1, 1, false, 0, nullptr, nullptr, false, &low, nullptr));
        drop(u);
        *out = Map{d->alloc((size_t)B * 4 * low.H * low.W * 256), 2 * low.H, 2 * low.W, 256};
        int rc = HIVE_OK;
        if (!dry) {
            const void *ob;
            DPT_TRY(need(pre + "out_conv.bias", &ob));
            rc = hive_nhwc_upsample2x(ctx, low.p, ob, dt, B, low.H, low.W, 256, out->p);
        }
        drop(low);
        return rc;
    };
    const Map *layers[4] = {&layer_1, &layer_2, &layer_3, &layer_4};
    Map path{}, prev{};
    for (int n = 4; n >= 1; --n) {
        Map lrn, lrn_relu;
        DPT_TRY(conv(*layers[n - 1], "scratch.layer" + std::to_string(n) + "_rn.weight", nullptr, 256, 3, 1, false, 0, nullptr, nullptr, true, &lrn, &lrn_relu));
        d->release(layers[n - 1]->p);  // layer_n fed its layer_rn convolution only
        DPT_TRY(refinenet(n, n == 4 ? nullptr : &prev, lrn, lrn_relu, &path));
        drop(lrn);
        drop(lrn_relu);
        if (n != 4) drop(prev);
        prev = path;
    }

    // ---- depth head: conv 256 -> 128 (+ bias, in its epilogue), x2 upsample + conv 128 -> 32 + ReLU + 1x1 + ReLU + inversion
    Map lo;
    DPT_TRY(conv(path, "scratch.output_conv.0.weight", "scratch.output_conv.0.bias", 128, 3, 1, false, 0, nullptr, nullptr, false, &lo, nullptr));
    drop(path);
    float *net_depth = resized ? (float *)d->alloc((size_t)B * H * W * 2) : nullptr;  // f32 depth at the network's size (2 half_t per float)
    if (dry) return HIVE_OK;
    const void *w3;
    DPT_TRY(need("scratch.output_conv.2.weight", &w3));  // [ky][kx][32][128]
    HIVE_REQUIRE(ctx, 2 * lo.H == H && 2 * lo.W == W, "hive_dpt: network size %d x %d must be a multiple of 32", H, W);
    if (!resized)
        return hive_dpt_head_fused(ctx, lo.p, nullptr, dt, B, lo.H, lo.W, 128, 32, w3, d->cfg.head_b3, d->cfg.head_w1, d->cfg.head_b1,
                                   d->cfg.non_negative, d->cfg.invert, d->cfg.scale, d->cfg.shift, d_depth, 1.0f / 1000.0f, max_depth, d_mm, d_m);
    // frames of another size: the head leaves float32 depth at the network's size, the nearest-neighbour resize back to the frame size
    // (dataset_adaptors.py:1421-1426) carries the uint16-mm hand-off
    DPT_TRY(hive_dpt_head_fused(ctx, lo.p, nullptr, dt, B, lo.H, lo.W, 128, 32, w3, d->cfg.head_b3, d->cfg.head_w1, d->cfg.head_b1, d->cfg.non_negative, d->cfg.invert,
                                d->cfg.scale, d->cfg.shift, net_depth, 1.0f / 1000.0f, max_depth, nullptr, nullptr));
    return hive_depth_resize_nearest(ctx, net_depth, B, H, W, FH, FW, 1.0f / 1000.0f, max_depth, d_depth, d_mm, d_m);
}

}  // namespace

extern "C" {

int hive_dpt_create(hive_ctx *ctx, const hive_dpt_config *config, const hive_dpt_tensor *tensors, int n_tensors, hive_dpt **out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, config && tensors && n_tensors > 0 && out, "hive_dpt_create: NULL argument");
    HIVE_REQUIRE(ctx, config->backbone == 0 || config->backbone == 1, "hive_dpt_create: backbone %d (0 = vitb_rn50_384, 1 = vitl16_384)", config->backbone);
    HIVE_REQUIRE(ctx, config->dtype == HIVE_BF16 || config->dtype == HIVE_F16, "hive_dpt_create: dtype %d (HIVE_F16 = %d or HIVE_BF16 = %d)", config->dtype, (int)HIVE_F16,
                 (int)HIVE_BF16);
    const int vit_depth = config->backbone == 0 ? 12 : 24, vit_dim = config->backbone == 0 ? 768 : 1024, vit_heads = vit_dim / 64;
    *out = nullptr;
    hive_dpt *d = new hive_dpt();
    d->ctx = ctx;
    d->cfg = *config;
    for (int i = 0; i < n_tensors; ++i)
        if (tensors[i].name && tensors[i].data) d->w[tensors[i].name] = tensors[i].data;
    // the ViT engine over the table's block weights
    std::vector<hive_vit_block_weights> blocks(vit_depth);
    const char *fields[12] = {"norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
                              "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"};
    for (int i = 0; i < vit_depth; ++i) {
        const void **dst = reinterpret_cast<const void **>(&blocks[i]);
        for (int f = 0; f < 12; ++f) {
            const std::string name = "pretrained.model.blocks." + std::to_string(i) + "." + fields[f];
            dst[f] = d->get(name);
            if (!dst[f]) {
                int rc = hive_fail(ctx, HIVE_ERR_INVALID, "hive_dpt_create: tensor '%s' missing from the table", name.c_str());
                delete d;
                return rc;
            }
        }
    }
    int rc = hive_vit_create(ctx, config->dtype, vit_depth, vit_dim, vit_heads, 4 * vit_dim, config->ln_eps, blocks.data(), &d->vit);
    if (rc) {
        hive_dpt_destroy(d);
        return rc;
    }
    *out = d;
    return HIVE_OK;
}

int hive_dpt_forward_frames(hive_dpt *d, const uint8_t *d_rgb, int B, int frame_h, int frame_w, int net_h, int net_w, const void *d_pos_embed, float *d_depth,
                            float max_depth, uint16_t *d_out_mm, float *d_out_m) {
    HIVE_ENTER(d ? d->ctx : nullptr);
    if (!d) return hive_fail(nullptr, HIVE_ERR_INVALID, "dpt is NULL");
    hive_ctx *ctx = d->ctx;
    const int H = net_h, W = net_w;
    HIVE_REQUIRE(ctx, d_rgb && d_pos_embed && (d_depth || d_out_mm || d_out_m), "hive_dpt_forward: NULL argument");
    HIVE_REQUIRE(ctx, B > 0 && frame_h > 0 && frame_w > 0, "hive_dpt_forward: bad frame batch %d x %d x %d", B, frame_h, frame_w);
    HIVE_REQUIRE(ctx, H >= 32 && W >= 32 && H % 32 == 0 && W % 32 == 0, "hive_dpt_forward: the network's input size must be a multiple of 32 (%d x %d)", H, W);
    // size the activation arena with a dry run of the same allocation sequence, then launch
    void *arena = d->arena;
    d->arena = nullptr;
    d->reset_arena();
    int rc = run_forward(d, true, d_rgb, B, frame_h, frame_w, H, W, d_pos_embed, d_depth, max_depth, d_out_mm, d_out_m);
    d->arena = arena;
    if (rc) return rc;
    const size_t need = d->arena_used;  // the high-water mark of the dry run
    if ((rc = hive_reserve_device(ctx, &d->arena, &d->arena_bytes, need))) return rc;
    d->reset_arena();
    return run_forward(d, false, d_rgb, B, frame_h, frame_w, H, W, d_pos_embed, d_depth, max_depth, d_out_mm, d_out_m);
}

int hive_dpt_forward(hive_dpt *d, const uint8_t *d_rgb, int B, int H, int W, const void *d_pos_embed, float *d_depth, float max_depth,
                     uint16_t *d_out_mm, float *d_out_m) {
    return hive_dpt_forward_frames(d, d_rgb, B, H, W, H, W, d_pos_embed, d_depth, max_depth, d_out_mm, d_out_m);
}

int hive_dpt_weights_modified(hive_dpt *d) {
    HIVE_ENTER(d ? d->ctx : nullptr);
    if (!d) return hive_fail(nullptr, HIVE_ERR_INVALID, "dpt is NULL");
    // everything derived from the caller's tensors: the ViT engine's folded LayerNorm weights (rebuilt now, on the context's stream) and the Gram tables of
    // the 1 x 1 convolutions (dropped; rebuilt by the forward that next needs them).  All other weights are read through the caller's pointers at every forward.
    HIVE_CHECK_HIP(d->ctx, hipStreamSynchronize(d->ctx->stream));  // (a forward in flight may still read the tables)
    for (auto &kv : d->gram_tables)
        if (kv.second) (void)hipFree(kv.second);
    d->gram_tables.clear();
    return hive_vit_weights_modified(d->vit);
}

int hive_dpt_arena_bytes(hive_dpt *d, int64_t *bytes) {
    if (!d || !bytes) return hive_fail(d ? d->ctx : nullptr, HIVE_ERR_INVALID, "hive_dpt_arena_bytes: NULL argument");
    *bytes = (int64_t)d->arena_bytes;
    return HIVE_OK;
}

int hive_dpt_destroy(hive_dpt *d) {
    if (!d) return HIVE_OK;
    {
        HIVE_ENTER(d->ctx);
        (void)hipStreamSynchronize(d->ctx->stream);
        if (d->vit) hive_vit_destroy(d->vit);
        if (d->arena) (void)hipFree(d->arena);
        for (auto &kv : d->gram_tables)
            if (kv.second) (void)hipFree(kv.second);
    }
    delete d;
    return HIVE_OK;
}

}  // extern "C"
