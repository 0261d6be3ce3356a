/*
 * gpp.h -- C ABI of libgpp_hip.so: the MI355X (gfx950) implementation of the data-parallel
 * hot path of arangesh/Ground-Plane-Polling, i.e. everything that one
 *     model.predict_on_batch([images, P_inv, planes])
 * (reference keras_retinanet_3D/bin/run_network.py:110) executes on the device:
 * RetinaNet-3D forward (ResNet + FPN + three heads), anchor decode, NMS / top-k, and the
 * per-detection ground-plane polling.
 *
 * The reference has no native code and no FFI; its "operator API" for this path is the
 * alias table keras_retinanet_3D/backend/tensorflow_backend.py:20-156 plus the Keras layers
 * in keras_retinanet_3D/layers/.  Each entry point below names the reference code it
 * replaces.  INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns int: 0 = GPP_OK, < 0 = argument error (below), > 0 = hipError_t
 *   - never throws, never aborts, allocates nothing: the caller owns every buffer, including
 *     workspaces whose sizes are reported by the *_workspace_bytes functions
 *   - all pointers are DEVICE pointers unless named host_*; kernels are enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream) and run asynchronously
 *   - no global state except per-device launch configuration and the side streams of gpp_plan_run lanes (both
 *     created on first use under a lock, one set per device); safe to call from several host threads on distinct streams
 *   - tensors are dense, row-major, NHWC for images / feature maps
 */
#ifndef GPP_H_
#define GPP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPP_OK 0
#define GPP_ERR_BAD_ARG (-1)    /* null pointer, negative size, unsupported shape */
#define GPP_ERR_WORKSPACE (-2)  /* workspace too small */
#define GPP_ERR_ALIGN (-3)      /* pointer not aligned as documented */
#define GPP_ERR_UNSUPPORTED (-4)

/* Library / build identification: "gpp-hip <version> gfx950". Host pointer, static storage. */
const char* gpp_version(void);

/* ------------------------------------------------------------------------------------------
 * Ground-plane polling.
 * Replaces layers/fit_road_planes.py:49-139 `fit_road_planes` (+ `poll` :18-32, `calc_X_t`
 * :34-47) and the `FitRoadPlanes` layer :142-186.
 *
 *   boxes      (B, D, 12) f32   x1 y1 x2 y2 xl yl xm ym xr yr xt yt   (-1 rows = padding)
 *   dims       (B, D, 3)  f32   h w l
 *   orient     (B, D)     i32   orientation class 0..3, -1 = padding
 *   P_inv      (B, 4, 3)  f32   pseudo-inverse of the scaled camera matrix
 *   planes     (N, 4) f32 if planes_batched == 0 (one database shared by the batch), else
 *              (B, N, 4) as the reference feeds it (preprocessing/kitti.py:220); 16-byte aligned
 *   thr        poll threshold in metres (reference constant 0.7, fit_road_planes.py:94)
 *   keypoints  (B, D, 4, 3) f32  X_l X_m X_r X_t on the selected plane
 *   keyplanes  (B, D, 1, 4) f32  the selected plane, canonicalised (normal up, unit norm)
 *   residuals  (B, D)     f32   masked residual of the selected plane / 6
 *   best_idx   (B, D)     i32   index of the selected plane (may be NULL; the reference
 *                               computes it at :119 but never returns it)
 *   workspace  gpp_poll_workspace_bytes() bytes, 16-byte aligned (canonical planes)
 * ---------------------------------------------------------------------------------------- */
int gpp_poll_workspace_bytes(int B, int N, int planes_batched, size_t* bytes);

int gpp_poll_f32(const float* boxes, const float* dims, const int32_t* orient, const float* P_inv,
                 const float* planes, int B, int D, int N, int planes_batched, float thr,
                 float* keypoints, float* keyplanes, float* residuals, int32_t* best_idx,
                 void* workspace, size_t workspace_bytes, void* stream);


/* ------------------------------------------------------------------------------------------
 * Element types of activations / weights of the convolution kernels.
 * ---------------------------------------------------------------------------------------- */
#define GPP_BF16 1   /* bfloat16 storage, float32 MFMA accumulation (default compute type) */
#define GPP_F16 2    /* IEEE half storage, float32 MFMA accumulation */
#define GPP_F32 3    /* float32 storage AND float32 operands (v_mfma_f32_16x16x4_f32: every product rounded once, float32
                        accumulation): the arithmetic type of the reference, keras.backend.floatx() = float32
                        (utils/image.py:47, placeholders models/retinanet.py:395-396).  1/16 of the 16-bit MFMA rate. */
#define GPP_BF16X3 4 /* float32 storage, each float32 product as three bf16 matrix products: x = hi + lo (hi = bf16(x), lo =
                        bf16(x - hi)), x*w ~ hi*whi + hi*wlo + lo*whi, float32 accumulation: ~2^-16 relative error per product
                        (float32: 2^-24, plain bf16 operands: 2^-8) at a third of the bf16 MFMA rate.  Activations, residuals
                        and outputs are float32 exactly as for GPP_F32; the weight matrix holds, per K-step of 32 input
                        channels, the 32 bf16 hi parts followed by the 32 bf16 lo parts (same bytes per row as float32). */
#define GPP_F16X3 5  /* as GPP_BF16X3 with IEEE-half halves: hi = f16(x), lo = f16(x - hi), 11 + 11 significant bits, ~2^-22 relative
                        error per product (float32: 2^-24): the throughput mode that stays inside the reference-precision tolerance
                        (plane index exact, 3-D corners within 1e-3 of the float32 path).  Range: finite activations beyond +-65504 are
                        clamped when they are split and COUNTED (gpp_x3_range_events below); a non-finite activation stays non-finite
                        (hi = x, lo = x - x: NaN or inf), as in the float32 path -- a broken activation is never laundered into a
                        plausible finite value; the packed weights of an output channel are scaled by a power of two so that
                        both halves are normal halfs, and gpp_conv_desc.out_scale (float32 per output channel, the inverse power of
                        two) is applied to the accumulator before the bias. */

/* ------------------------------------------------------------------------------------------
 * 2-D convolution, NHWC, implicit GEMM on MFMA (no im2col buffer), fused epilogue
 *     out = act( conv(in, weight) + bias [+ nearest_resize(residual)] )
 * Replaces every Conv2D (+ frozen BatchNormalization folded into weight/bias, + ReLU, + Add,
 * + UpsampleLike) of the graph built by models/retinanet.py:24-205 (heads :24-167, FPN
 * :170-205; UpsampleLike layers/_misc.py:90-100) and of the third-party keras_resnet
 * bottleneck stack instantiated at models/resnet.py:88-93, except the 3-channel stem.
 *
 * One launch covers up to GPP_MAX_GROUPS independent feature maps that share the weights
 * (the five pyramid levels of a head layer, retinanet.py:257-281), each described by a
 * gpp_conv_group.  GEMM view per group: M = batch*H_out*W_out output pixels, N = C_out,
 * K = KH*KW*C_in, K ordered (c_in / CK, kh, kw, c_in % CK) with CK = 128 bytes of channels (64 for the 16-bit types,
 * 32 for the float32-sized types GPP_F32 / GPP_BF16X3 / GPP_F16X3): the taps of one channel chunk are adjacent so that their
 * overlapping input rows are re-read from the XCD-local L2.
 *
 * Layouts (element = 2 bytes for GPP_BF16 / GPP_F16, 4 bytes for GPP_F32 / GPP_BF16X3 / GPP_F16X3; a pre-split x3 map, see x3_split,
 * has the same 4 bytes per element: 32 channels = 64 bytes of hi halves + 64 bytes of lo halves)
 *   in        pixel (b, y, x) of a group at  in + in_off + b*in_bstride + (y*W_in + x)*in_pitch,
 *             C_in contiguous channels there (in_pitch >= C_in lets a channel slice be read)
 *   weight    [C_out rounded up to a multiple of 256][KH*KW*C_in], K contiguous; rows >= C_out
 *             must exist (zero) -- weight_rows states how many rows are allocated.  Rows are
 *             interleaved within every group of 32 output channels: stored row 16h + 4q + r holds
 *             output channel 8q + 4h + r (h in 0..1, q, r in 0..3), so that a lane of the MFMA result
 *             owns 8 consecutive output channels (16-byte stores straight from the accumulators)
 *   bias      [C_out] float32 (NULL = none)
 *   residual  same addressing as out with res_* fields; when H_res/W_res differ from
 *             H_out/W_out the residual is read with TF nearest-neighbour resize semantics
 *             src = min(floor(dst * in/out), in-1)  (tf.image.resize_images, align_corners=False)
 *   out       element type = dtype, or float32 when out_f32 != 0
 *   zero_page unused since v0.2 (padding comes from range-checked buffer loads); may be NULL
 * Requirements: C_in % CK == 0; C_out % 4 == 0; in_pitch, out_pitch, res_pitch multiples of 16 bytes
 * (8 elements; 4 for float32); all base pointers 16-byte aligned; stride in {1, 2}.
 * Padding is explicit (pad_top, pad_left); bottom/right padding is implied by H_out/W_out
 * (this covers Keras 'same' at stride 1, TF's asymmetric 'same' at stride 2, and
 * ZeroPadding2D + 'valid').  Every input map and the weight tensor must be smaller than 2 GiB.
 * ---------------------------------------------------------------------------------------- */
#define GPP_MAX_GROUPS 5

typedef struct gpp_conv_group {
    int64_t in_off, in_bstride;     /* elements */
    int64_t out_off, out_bstride;
    int64_t res_off, res_bstride;
    int32_t H_in, W_in, H_out, W_out;
    int32_t H_res, W_res;
    int32_t tile_start;             /* filled in by the library */
    int32_t row_begin;              /* filled in by the library (first GEMM row of the group this grid part covers; callers leave 0) */
} gpp_conv_group;

typedef struct gpp_conv_desc {
    const void* in;
    const void* weight;
    const float* bias;
    const void* residual;
    void* out;
    const void* zero_page;
    int32_t dtype;                  /* GPP_BF16 | GPP_F16 | GPP_F32 | GPP_BF16X3 | GPP_F16X3 */
    int32_t out_f32;
    int32_t batch, C_in, C_out, KH, KW, stride, pad_top, pad_left;
    int32_t in_pitch, out_pitch, res_pitch;   /* elements per pixel */
    int32_t weight_rows;
    int32_t relu;
    int32_t n_groups;
    int32_t tile_hint;              /* 0 = library heuristic; BM*1000 + BN forces a block tile (64..224 x 64/128, 128/192 x 160,
                                       256256), + 1000000 = the software-pipelined main loop (128128, 192128, 128256, 192256,
                                       192160, 128160, 192096), 2256256 = 256256 plus 512 x 128 tiles for the last 128 columns in one grid
                                       (C_out = 256 k + 128 only); legacy codes 64 / 128 / 256 / 512; anything else: GPP_ERR_BAD_ARG.
                                       GPP_BF16X3 / GPP_F16X3 (the same tile set, the same loops): the plain tiles, 128256 / 192256 / 256256
                                       (8 wavefronts), and on a pre-split input map (x3_split & GPP_X3_IN) the pipelined 1128128, 1192128,
                                       1128256, 1160256, 1192256, 1224256, 1256256, 1128160, 1192096 and 2256256; 192160 exists on pre-split inputs only;
                                       3256224 / 3192160 (3000000 + BMA * 1000 + BMB, C_out % 256 == 0, pre-split inputs): 256-column tiles
                                       of two heights in ONE grid -- whole rounds of BMA-row tiles, the rest in BMB-row tiles -- against the
                                       round quantisation of one-workgroup-per-CU tiles (GPP_ERR_UNSUPPORTED where it gains nothing);
                                       4128064 / 4064064 / 4128128 / 4064128 (4000000 + BM * 1000 + BN; 1 x 1, stride 1, one map, pre-split input / shortcut,
                                       C_out a multiple of BN with 32 % (C_out / BN) == 0, K small enough for BN x K weights + the activation ring in 160 KB of LDS): the
                                       weight-stationary persistent form of the shallow 1 x 1 layers;
                                       5064064 / 5096064 / 5064128 / 5096128 / 5128128 (5000000 + BM * 1000 + BN; x3 types, pre-split inputs): the plain loop on a
                                       four-deep LDS ring -- for launches of at most about one workgroup per CU (deep-K small-M layers, batch 1), whose
                                       K-steps are bound by the latency of their own tile loads.
                                       gpp_conv2d_tile_candidates lists what a given layer accepts; see gpp_conv2d_autotune */
    int32_t reserved;               /* must be 0 (anything else: GPP_ERR_BAD_ARG).  Only the diagnostic -DGPP_STAMPS build of the
                                       library (make stamps; tools/bench_conv.py) reads it: bit 0 skip the tile loads, bit 1 skip
                                       the LDS reads + MFMA, bit 2 / 3 flip the pipelined form of the 128128 / 256256 tile,
                                       bits 4, 5 enable in-kernel time stamps (written through zero_page) */
    int32_t in_bytes, weight_bytes; /* filled in by the library: extents for the range-checked buffer loads */
    void* partial;                  /* optional split-K workspace (float32 partial tiles), 16-byte aligned; NULL = never split */
    int64_t partial_bytes;          /* >= gpp_conv2d_workspace_bytes(), else GPP_ERR_WORKSPACE when the layer is split */
    int32_t split_k;                /* 0 = gpp_conv2d_split_rule (a function of the layer alone), 1 = never, k > 1 = exactly k */
    int32_t partial_rows;           /* filled in by the library */
    gpp_conv_group groups[GPP_MAX_GROUPS];
    int32_t x3_split;               /* GPP_BF16X3 / GPP_F16X3 only (0 otherwise): which of the float32-sized maps hold PRE-SPLIT values, bits
                                       GPP_X3_IN | GPP_X3_OUT | GPP_X3_RES.  A pre-split map stores every 32 channels of a pixel
                                       (128 bytes) as [32 bf16 hi | 32 bf16 lo], hi = bf16(x), lo = bf16(x - hi) -- the layout the
                                       packed weights already have -- instead of 32 float32: the matrix loop then takes its
                                       operands straight from LDS, without the per-fragment split on the vector ALU that shares
                                       issue slots with the matrix pipe.  Needs pitches and offsets that are multiples of 32
                                       channels; an output map can be pre-split only when out_f32 == 0 and C_out % 32 == 0 */
    int32_t reserved2;              /* must be 0 */
    const float* out_scale;         /* GPP_F16X3 only (NULL otherwise, and NULL = all ones): per output channel, accumulator *= out_scale[n]
                                       before the bias -- the inverse of the power of two the channel's packed weights were scaled by */
    uint64_t* range_counter;        /* GPP_F16X3 only: device address of the 8-byte counter this launch adds its range events to (see
                                       gpp_x3_range_events below) -- a caller that runs several models or streams gives each plan a slot of
                                       its own and reads THAT (gpp_x3_range_snapshot_of); NULL = the library's per-device counter */
} gpp_conv_desc;
#define GPP_X3_IN 1
#define GPP_X3_OUT 2
#define GPP_X3_RES 4

int gpp_conv2d_igemm(const gpp_conv_desc* host_desc, void* stream);

/* Split-K factor the library uses for this layer when desc->split_k == 0 and a workspace is given.  It is a function of
   the layer alone (kernel size, channels, output pixels PER IMAGE) -- never of the batch size, the block tile or a timing --
   so the float32 summation order of every output element, hence every bit of a result, is the same whether an image is
   computed alone, inside a larger batch or on another rank.  gpp_conv2d_workspace_bytes: size of `partial` that any
   block tile of this layer may need (0 when the layer is not split). */
int gpp_conv2d_split_rule(const gpp_conv_desc* host_desc, int* split_k);
int gpp_conv2d_workspace_bytes(const gpp_conv_desc* host_desc, size_t* bytes);

/* Time the block-tile candidates of this layer on the device (iters launches each; the layer only rewrites its own
   output) and store the fastest in desc->tile_hint.  best_us (optional): its time per launch.  Synchronises the stream.
   Results do not depend on the tile (same K order per output element); split_k is used as given, never tuned. */
int gpp_conv2d_autotune(gpp_conv_desc* desc, int iters, void* stream, float* best_us);

/* The tile codes gpp_conv2d_autotune would time for this layer (count = how many there are; the first min(count, capacity)
   are written to tiles).  Any of them gives the same bytes; tests draw from this list at random (GPP_TUNE_RANDOM). */
int gpp_conv2d_tile_candidates(const gpp_conv_desc* host_desc, int* tiles, int capacity, int* count);

/* Fused tail of a ResNet bottleneck (keras_resnet bottleneck_2d, used at /root/reference/keras_retinanet_3D/models/
   resnet.py:88-93): the 3x3 conv "branch2b" (C -> C, C = 64 or 128, stride 1, pad 1, + bias + ReLU) and the 1x1 conv
   "branch2c" (C -> multiple of 128, + bias + residual + ReLU) in ONE launch; the intermediate map stays in LDS.
   conv3x3->out is not written.  Results are bit-identical to gpp_conv2d_igemm(conv3x3) + gpp_conv2d_igemm(conv1x1).
   tile_rows: 0 (= 128), 96, 128 or 160 output pixels per workgroup (the x3 types also 64: three workgroups per CU).  Other shapes,
   and GPP_F32: GPP_ERR_UNSUPPORTED. */
int gpp_bottleneck_tail(const gpp_conv_desc* conv3x3, const gpp_conv_desc* conv1x1, int tile_rows, void* stream);

/* A WHOLE bottleneck of the same graph in one launch (GPP_F16X3 / GPP_BF16X3 on pre-split maps): "branch2a" (1x1, C_in -> C, stride 1 or 2, + bias
   + ReLU), "branch2b" (3x3, C -> C, stride 1, pad 1, + bias + ReLU) and "branch2c" (1x1, C -> multiple of 128, + bias + shortcut + ReLU), C = 64 or
   128; both intermediate maps stay in LDS (a workgroup computes a tile of 8 x 14 output pixels and recomputes branch2a on its one-pixel halo).
   conv1x1_a->out and conv3x3_b->out are not written.  Results are bit-identical to the three gpp_conv2d_igemm launches (same K order and
   the same epilogue arithmetic per output element; range events counted once per stored group, as the three launches count them).  The
   shortcut is conv1x1_c->residual: a map a projection launch wrote, or the block's own input map -- an identity block (same pointer, offsets and
   pitches as conv1x1_a's input, stride 1, C_in = 4 C), whose shortcut rows are then taken from the LDS ring they pass through anyway and the map is
   read once (C = 64) / 1.25 times (C = 128) instead of twice.  tile: 0 = the library's choice (814 = 8 x 14 pixels, the only tile built);
   + 1000 forces the general form (the shortcut read from its map) on an identity block; + 10000 k: the odd tile rows start k microseconds late
   (an experiment).  Every map of the block must stay below 2 GiB per image.  Other shapes / types: GPP_ERR_UNSUPPORTED. */
int gpp_bottleneck_block(const gpp_conv_desc* conv1x1_a, const gpp_conv_desc* conv3x3_b, const gpp_conv_desc* conv1x1_c, int tile, void* stream);

/* GPP_F16X3 range ledger.  The half type ends at +-65504: an epilogue that stores an activation outside it (a finite value it has to
   clamp, an inf or a NaN) adds one event per 8-channel group to a device-side counter (one per device).  host_count (host pointer,
   may be NULL) receives the events since the last reset; reset != 0 clears the counter.  Synchronises the device.  Zero after a run
   means no activation of that run was altered by the type's range (models/retinanet.py: model.x3_range_events()). */
int gpp_x3_range_events(uint64_t* host_count, int reset);

/* The same counter WITHOUT a synchronisation: one tiny launch on `stream` copies the counter's value at that point of the stream
   into *device_count (8 bytes of device memory, or of page-locked host memory the device can write).  Enqueued behind a plan run,
   the value tells -- once the stream has reached it -- whether an activation of that run (or of any earlier one since the last
   reset) left the half range; models/retinanet.py reads it with the results it fetches anyway and re-runs an affected call at
   float32 (`on_range_event`), so that a clamped activation is never returned as a plausible wrong answer. */
int gpp_x3_range_snapshot(uint64_t* device_count, void* stream);

/* The same for a counter of the caller's (the slot its descriptors name in gpp_conv_desc.range_counter / gpp_stem_desc.range_counter; NULL = the
   library's per-device counter).  A caller that runs several GPP_F16X3 models, or one model from several streams, gives every plan its own
   8-byte slot: what one plan's launches count is then invisible to every other plan -- no spurious reaction, no missed one, whoever resets
   what (models/retinanet.py: Plan.range_slot). */
int gpp_x3_range_snapshot_of(const uint64_t* counter, uint64_t* device_count, void* stream);

/* Algorithmic FLOPs (2 * MACs) of one launch described by host_desc. */
int gpp_conv2d_flops(const gpp_conv_desc* host_desc, double* flops);

/* ------------------------------------------------------------------------------------------
 * ResNet stem and small element-wise helpers.
 * gpp_stem_conv7x7_bn_relu replaces keras_resnet's ZeroPadding2D(3) + conv1 (7x7, stride 2,
 * no bias) + bn_conv1 (frozen, eps 1e-5) + ReLU (instantiated at models/resnet.py:88-93):
 *   in (B, H, W, 3) float32 BGR mean-subtracted (utils/image.py:36-62), weight [7*7*3][64]
 *   float32 = Keras HWIO kernel with the BN scale folded in, bias [64] = folded BN shift,
 *   out (B, Ho, Wo, 64) of `dtype`, Ho = (H + 6 - 7)/2 + 1.  Float32 fmaf chain on the vector ALUs: the stem of the
 *   GPP_F32 (reference-precision) path; the 16-bit paths use the MFMA form below.
 * gpp_maxpool3x3s2_same replaces MaxPooling2D(3x3, stride 2, padding 'same') 'pool1'.
 * gpp_relu replaces Activation('relu') 'C6_relu' (models/retinanet.py:202).
 * ---------------------------------------------------------------------------------------- */
int gpp_stem_conv7x7_bn_relu(const float* in, const float* weight, const float* bias, void* out, int dtype,
                             int B, int H, int W, void* stream);
/* MFMA form of the stem (the one the model uses): float32 input and the packed weights are rounded to f16
 * (11-bit significand) on the fly, accumulated in float32 by v_mfma_f32_16x16x32_f16.  packed_weight_f16 is
 * the [64][232] f16 image that the HOST-side helper gpp_stem_pack_weights_f16 produces from the [147][64]
 * float32 folded kernel (14 848 elements = 29 696 bytes; both pointers of the helper are host pointers). */
int gpp_stem_pack_weights_f16(const float* host_weight_147x64, void* host_packed, size_t packed_bytes);
int gpp_stem_conv7x7_bn_relu_mfma(const float* in, const void* packed_weight_f16, const float* bias, void* out,
                                  int dtype, int B, int H, int W, void* stream);
int gpp_maxpool3x3s2_same(const void* in, void* out, int dtype, int B, int H, int W, int C, void* stream);
/* conv1 + bn_conv1 + ReLU + pool1 in ONE launch (GPP_BF16 / GPP_F16): out is the POOLED map (B, Hp, Wp, 64), Hp = (Ho + 1)/2;
 * the (B, Ho, Wo, 64) conv map is never written (137 MB at B = 8, 402 x 1333).  Bit-identical to
 * gpp_stem_conv7x7_bn_relu_mfma followed by gpp_maxpool3x3s2_same (the max is taken over the rounded conv values). */
int gpp_stem_pool_fused_mfma(const float* in, const void* packed_weight_f16, const float* bias, void* out,
                             int dtype, int B, int H, int W, void* stream);
/* The stem of the float32-storage "x3" types (GPP_F16X3 / GPP_BF16X3 models): conv1 + bn_conv1 + ReLU on the matrix pipe at
 * (almost) float32 precision -- input pixels and weights split into two IEEE halves each, three matrix products per float32 product,
 * float32 output (B, Ho, Wo, 64).  packed_weight_x3 = what the HOST-side helper gpp_stem_pack_weights_f16x3 writes from the folded
 * [147][64] kernel: 2 * 64 * 232 halfs + 64 float32 (59 648 bytes). */
int gpp_stem_pack_weights_f16x3(const float* host_weight_147x64, void* host_packed, size_t packed_bytes);
int gpp_stem_conv7x7_bn_relu_x3(const float* in, const void* packed_weight_x3, const float* bias, float* out,
                                int B, int H, int W, void* stream);
/* ... counting its range events (output values the half type cannot hold: the map is split -- and clamped -- by the layers that read it) into
 * the caller's 8-byte slot instead of the library's per-device counter (NULL = that one: the function above) */
int gpp_stem_conv7x7_bn_relu_x3_rc(const float* in, const void* packed_weight_x3, const float* bias, float* out,
                                   int B, int H, int W, uint64_t* range_counter, void* stream);
/* conv1 + bn_conv1 + ReLU + pool1 of the x3 types in ONE launch: out is the POOLED float32 map (B, Hp, Wp, 64); the (B, Ho, Wo, 64) float32
 * conv map (274 MB at B = 8, 402 x 1333) is never written.  Bit-identical to gpp_stem_conv7x7_bn_relu_x3_rc followed by
 * gpp_maxpool3x3s2_same(GPP_F32), and the range events of the conv map are counted as there (same count). */
int gpp_stem_pool_fused_x3(const float* in, const void* packed_weight_x3, const float* bias, float* out,
                           int B, int H, int W, uint64_t* range_counter, void* stream);
/* dtype GPP_BF16X3 = a pre-split map (gpp_conv_desc.x3_split): ReLU on the [hi | lo] pairs (count in float32-sized elements, a
   multiple of 32) */
int gpp_relu(const void* in, void* out, int dtype, int64_t count, void* stream);
/* batched form: image b reads `count` elements at in + b*in_bstride, writes out + b*out_bstride */
int gpp_relu_strided(const void* in, int64_t in_bstride, void* out, int64_t out_bstride, int dtype, int B,
                     int64_t count, void* stream);

/* ------------------------------------------------------------------------------------------
 * Image preprocessing: uint8 BGR frames (B, H, W, 3) -> float32 (B, Ho, Wo, 3),
 * ImageNet mean subtracted per channel, then bilinear resize.  Replaces the host-side
 * utils/image.py:36-62 (preprocess_image) + :174-200 (resize_image -> cv2.resize INTER_LINEAR).
 * y0/y1/wy (Ho entries) and x0/x1/wx (Wo entries) are the interpolation taps: source indices and the
 * float32 weight of the second one, as cv2 derives them from the scale (half-pixel centres, border
 * replicated); device arrays, computed once per input shape by the host.
 * ---------------------------------------------------------------------------------------- */
int gpp_preprocess_u8_bgr(const uint8_t* frames, float* out, const int32_t* y0, const int32_t* y1, const float* wy,
                          const int32_t* x0, const int32_t* x1, const float* wx, int B, int H, int W, int Ho, int Wo,
                          float mean_b, float mean_g, float mean_r, void* stream);

/* ------------------------------------------------------------------------------------------
 * Detection decode: sigmoid, orientation fold, score threshold, NMS, top-k, box / dimension
 * decode, -1 padding.  Replaces models/retinanet.py:72-73 (sigmoid), layers/_misc.py:133-141
 * + backend/common.py:43-81 (RegressBoxes), layers/_misc.py:186-187 + backend/common.py:23-40
 * (RegressDims) and layers/filter_detections.py:18-189 on its default path (nms=True,
 * class_specific_filter=True, orientation_specific_filter=False, one class).
 *
 *   cls_logits     (B, n_anchors, 8)  f32  pre-sigmoid classification head output
 *   regression     fused_layout == 0: (B, n_anchors, 12) f32 as the reference concatenates it
 *                  (retinanet.py:112-124); fused_layout == 1: (B, n_anchors/A, 12*A) f32, per
 *                  pixel [op1: 4A | op2: 2A | op3: 2A | op4: 2A | op5: 2A] (one fused conv)
 *   regression_dim (B, n_anchors, 3)  f32
 *   anchors        (n_anchors, 4)     f32  x1 y1 x2 y2 (layers/_misc.py:24-87), 16-byte aligned
 *   boxes (B, max_det, 12) dims (B, max_det, 3) scores (B, max_det) f32,
 *   labels / orientations (B, max_det) i32; rows past the survivors are -1
 *   anchor_index   (B, max_det) i32 anchor id of each detection (may be NULL; not a reference output)
 *   counts         (B) i32 number of anchors above score_thr per image (may be NULL)
 *   workspace      gpp_detect_workspace_bytes() bytes, 16-byte aligned
 * Reference constants: score_thr 0.05, iou_thr 0.5, max_det 100 (filter_detections.py:26-28).
 * ---------------------------------------------------------------------------------------- */
int gpp_detect_workspace_bytes(int B, int64_t n_anchors, size_t* bytes);

int gpp_detect_f32(const float* cls_logits, const float* regression, const float* regression_dim,
                   const float* anchors, int B, int64_t n_anchors, int num_base_anchors, int fused_layout,
                   float score_thr, float iou_thr, int max_det,
                   float* boxes, float* dims, float* scores, int32_t* labels, int32_t* orientations,
                   int32_t* anchor_index, int32_t* counts,
                   void* workspace, size_t workspace_bytes, void* stream);

/* The same work as three separately enqueueable stages (bit mask; gpp_detect_f32 = all three, in this order):
 *   GPP_DETECT_CANDIDATES  sigmoid + fold + threshold -> candidate keys      reads cls_logits only
 *   GPP_DETECT_SELECT      sort + greedy NMS -> survivors' keys              reads the keys + corner regressions
 *   GPP_DETECT_EMIT        full decode of the survivors, -1 padding           reads every head tensor
 * so that a caller can start the (latency-bound, one workgroup per image) selection as soon as the classification
 * and regression heads are done and overlap it with the dimension head (models/retinanet.py:128-167).  All stages
 * take the full argument list; state passes through `workspace`. */
#define GPP_DETECT_CANDIDATES 1
#define GPP_DETECT_SELECT 2
#define GPP_DETECT_EMIT 4
int gpp_detect_stages_f32(int stages, const float* cls_logits, const float* regression, const float* regression_dim,
                          const float* anchors, int B, int64_t n_anchors, int num_base_anchors, int fused_layout,
                          float score_thr, float iou_thr, int max_det,
                          float* boxes, float* dims, float* scores, int32_t* labels, int32_t* orientations,
                          int32_t* anchor_index, int32_t* counts,
                          void* workspace, size_t workspace_bytes, void* stream);

/* orientation_specific_filter=True (layers/filter_detections.py:84-98, a non-default argument of models.load_model): threshold
 * and NMS once per orientation on that orientation's folded score, the four survivor lists concatenated in orientation
 * order, then the common top-k; an anchor may be reported once per orientation.  Same arguments as gpp_detect_f32; its own
 * (4x larger) workspace; at most 16 images per call. */
int gpp_detect_osf_workspace_bytes(int B, int64_t n_anchors, size_t* bytes);
int gpp_detect_osf_f32(const float* cls_logits, const float* regression, const float* regression_dim,
                       const float* anchors, int B, int64_t n_anchors, int num_base_anchors, int fused_layout,
                       float score_thr, float iou_thr, int max_det,
                       float* boxes, float* dims, float* scores, int32_t* labels, int32_t* orientations,
                       int32_t* anchor_index, int32_t* counts,
                       void* workspace, size_t workspace_bytes, void* stream);

/* The eight result arrays -> one (B, D, 35) float32 tensor [12 box | 3 dim | score | label | orientation | 12 keypoints |
 * 4 plane | residual]: the unit of the per-step exchange between the ranks of a node (one all-gather). */
int gpp_pack_detections(const float* boxes, const float* dims, const float* scores, const int32_t* labels,
                        const int32_t* orientations, const float* keypoints, const float* keyplanes,
                        const float* residuals, int B, int D, float* packed, void* stream);

/* ------------------------------------------------------------------------------------------
 * Plan execution: one call enqueues a whole predict_on_batch (every kernel of the graph that
 * models/retinanet.py:359-422 `retinanet_bbox` builds) from a host array of descriptors.
 * The runner holds no state: the caller (Python) keeps the descriptors and buffers alive.
 * Ops whose `tag` is non-zero are bracketed by HIP events when `events` is given: the k-th
 * tagged op records events[2k] before and events[2k+1] after its launch, on `stream`
 * (this is how bench.py measures the dominant kernel live inside the timed region).
 * ---------------------------------------------------------------------------------------- */
#define GPP_OP_STEM 1
#define GPP_OP_MAXPOOL 2
#define GPP_OP_CONV 3
#define GPP_OP_RELU 4
#define GPP_OP_DETECT 5
#define GPP_OP_POLL 6
#define GPP_OP_BOTTLENECK_TAIL 7
#define GPP_OP_DETECT_CANDIDATES 8   /* gpp_detect_desc; stages of GPP_OP_DETECT, see gpp_detect_stages_f32 */
#define GPP_OP_DETECT_SELECT 9
#define GPP_OP_DETECT_EMIT 10
/* (11: the three-layer tail of round 2, removed in round 3 -- measured slower than its separate launches) */
#define GPP_OP_DETECT_OSF 12             /* gpp_detect_desc -> gpp_detect_osf_f32 */
#define GPP_OP_STEM_POOL 13              /* gpp_stem_desc with out = the pooled map -> gpp_stem_pool_fused_mfma (GPP_BF16 / GPP_F16) / gpp_stem_pool_fused_x3 (GPP_F16X3 / GPP_BF16X3) */
#define GPP_OP_BOTTLENECK_BLOCK 16       /* gpp_block_desc -> gpp_bottleneck_block */
/* (14, 15: the Winograd F(2, 3) form of the tower layers of round 5 -- built, measured at -2 % of the step, shelved in round 6:
   tools/experiments/winograd/) */
/* Optional concurrency inside a plan: `kind | GPP_OP_LANE(l)` (l = 1, 2) enqueues the op on a library-owned side stream
   that forks from the caller's stream at the first op of that lane; `kind | GPP_OP_JOIN` on a lane-0 op makes it wait
   for every open lane (the end of the plan joins too, and so does every error return: a failed gpp_plan_run leaves no forked work
   un-joined).  GPP_OP_JOIN on a side-lane op is GPP_ERR_BAD_ARG.  The caller orders the ops so that each lane only depends on
   what was enqueued before its fork.  The side streams belong to the DEVICE (one pair per device ordinal, made on first use): plans
   enqueued on different caller streams of one device share them, so their side-lane work is ordered plan after plan.
   Used for the projection shortcuts, the half-batch chains of res3-res5, the small FPN launches and the detection selection. */
#define GPP_OP_LANE(l) ((l) << 8)
#define GPP_OP_JOIN 0x10000
/* on a side-lane op: the lane first waits for everything enqueued on the caller's stream so far (a second fork point) */
#define GPP_OP_SYNC 0x20000

/* Optional stage label of an op, bits 20-23 of `kind`: with GPP_ROCTX=1 in the environment gpp_plan_run opens a roctx range ("gpp:stem", "gpp:backbone",
   "gpp:fpn", "gpp:heads", "gpp:decode", "gpp:polling") around each run of consecutive ops with the same label (rocprofv3 --marker-trace); 0 = none.
   The marker library is looked up at run time; without the variable nothing is loaded.  GPP_ROCTX=2 additionally synchronises the device where a range opens and
   closes: a range's duration in the marker trace is then its stage's time on the device (tools/roctx_stages.sh); a measuring mode, not a production one. */
#define GPP_OP_STAGE(s) (((s) & 15) << 20)
#define GPP_STAGE_STEM 1
#define GPP_STAGE_BACKBONE 2
#define GPP_STAGE_FPN 3
#define GPP_STAGE_HEADS 4
#define GPP_STAGE_DECODE 5
#define GPP_STAGE_POLLING 6

typedef struct gpp_stem_desc { const float* in; const void* weight; const float* bias; void* out;
                               int32_t dtype, B, H, W; uint64_t* range_counter; /* GPP_F16X3: see gpp_stem_conv7x7_bn_relu_x3_rc; NULL otherwise */
                             } gpp_stem_desc;   /* weight: packed f16 image (MFMA stem); GPP_F32: float32 [147][64]; GPP_F16X3: gpp_stem_pack_weights_f16x3 */
typedef struct gpp_pool_desc { const void* in; void* out; int32_t dtype, B, H, W, C, reserved; } gpp_pool_desc;
typedef struct gpp_relu_desc { const void* in; void* out; int64_t in_bstride, out_bstride, count;
                               int32_t dtype, B; } gpp_relu_desc;
typedef struct gpp_detect_desc {
    const float* cls_logits; const float* regression; const float* regression_dim; const float* anchors;
    float* boxes; float* dims; float* scores; int32_t* labels; int32_t* orientations;
    int32_t* anchor_index; int32_t* counts; void* workspace;
    size_t workspace_bytes; int64_t n_anchors;
    int32_t B, num_base_anchors, fused_layout, max_det;
    float score_thr, iou_thr;
} gpp_detect_desc;
typedef struct gpp_poll_desc {
    const float* boxes; const float* dims; const int32_t* orient; const float* P_inv; const float* planes;
    float* keypoints; float* keyplanes; float* residuals; int32_t* best_idx; void* workspace;
    size_t workspace_bytes;
    int32_t B, D, N, planes_batched;
    float thr; int32_t reserved;
} gpp_poll_desc;

typedef struct gpp_tail_desc { const gpp_conv_desc* conv3x3; const gpp_conv_desc* conv1x1; int32_t tile_rows, reserved; } gpp_tail_desc;
typedef struct gpp_block_desc { const gpp_conv_desc* conv1x1_a; const gpp_conv_desc* conv3x3_b; const gpp_conv_desc* conv1x1_c; int32_t tile, reserved; } gpp_block_desc;

typedef struct gpp_plan_op { int32_t kind; int32_t tag; const void* desc; } gpp_plan_op;

int gpp_plan_run(const gpp_plan_op* host_ops, int n_ops, void* stream, void* const* events, int n_events);

/* HIP events for callers without HIP headers (bench.py): timing enabled, host handles. */
int gpp_event_create(void** event);
int gpp_event_destroy(void* event);
int gpp_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on `stop` */

#ifdef __cplusplus
}
#endif

#endif /* GPP_H_ */
