// Plan runner: enqueue a whole predict_on_batch from a host array of descriptors (include/gpp.h).
// Stateless; every op forwards to the typed C-ABI entry point, so a plan run is exactly the same
// launches as calling those entry points one by one from the host language.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "gpp.h"

namespace {

// Side lanes: ops whose kind carries a lane number (kind | GPP_OP_LANE(lane), lane 1..kLanes) run on a
// library-owned stream that forks from the caller's stream where the lane is first used; a lane-0 op
// marked GPP_OP_JOIN (and the end of the plan) waits for all open lanes.  Independent chains of the graph (the three head
// towers) overlap their ramp-up / tail phases this way; results do not change.
// The side streams and their events belong to ONE device: there is a set per device ordinal, created on first use under
// a lock, and a plan that uses lanes holds that device's lock while it enqueues (two host threads driving the same
// device would otherwise interleave their fork / join events on the shared side streams).  Plans without lanes take no lock.
// Consequence (by design, documented in include/gpp.h): the lanes are per DEVICE, not per caller stream -- two plans enqueued on
// different user streams of one device (utils/pipeline.FramePipeline, tools/two_in_flight.py) share the same two side streams, so
// their side-lane work is ordered one plan after the other; their main-stream work still overlaps.
constexpr int kLanes = 2;
constexpr int kMaxDevices = 64;
struct Lanes {
    std::mutex lock;
    hipStream_t stream[kLanes] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, done[kLanes] = {nullptr, nullptr};
    bool ready = false;
    int init()                                          // call with `lock` held
    {
        if (ready) return GPP_OK;
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        // the side lanes run at the caller's (normal) priority: with half-batch chains and the FPN launches on them they are no longer
        // only "short latency-bound chains", and letting their workgroups in first cost 0.6 % of the f16x3 step (808 -> 813 images/s,
        // same box, alternating); GPP_LANE_PRIORITY=high restores the highest priority (read once, when the device's lanes are made)
        const char* pr = getenv("GPP_LANE_PRIORITY");
        const int prio = (pr && !strcmp(pr, "high")) ? hi : (pr && !strcmp(pr, "low")) ? lo : 0;
        for (int l = 0; l < kLanes; ++l) {
            hipError_t e = hipStreamCreateWithPriority(&stream[l], hipStreamNonBlocking, prio);
            if (e != hipSuccess) return (int)e;
            e = hipEventCreateWithFlags(&done[l], hipEventDisableTiming);
            if (e != hipSuccess) return (int)e;
        }
        hipError_t e = hipEventCreateWithFlags(&fork, hipEventDisableTiming);
        if (e != hipSuccess) return (int)e;
        ready = true;
        return GPP_OK;
    }
};
Lanes g_lanes_of_device[kMaxDevices];

// GPP_ROCTX=1: a roctx range per stage of the plan (GPP_OP_STAGE: stem, backbone, FPN, heads, decode, polling) around the launches that
// gpp_plan_run enqueues for it, so that `rocprofv3 --marker-trace --kernel-trace` gives the per-stage split of a step without
// matching kernel names.  The marker library is looked up at run time (librocprofiler-sdk-roctx.so, the one rocprofv3 listens to; libroctx64.so
// as a fallback): the library has no link-time dependency on it, and without the variable nothing is loaded and a plan run pays one branch.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool sync = false;       // GPP_ROCTX=2: the device is synchronised where a range opens and closes, so that a range's host duration IS its stage's time on the device
    Roctx()
    {
        const char* e = getenv("GPP_ROCTX");
        if (!e || (e[0] != '1' && e[0] != '2')) return;
        sync = e[0] == '2';
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};
const char* const kStageNames[16] = {nullptr, "gpp:stem", "gpp:backbone", "gpp:fpn", "gpp:heads", "gpp:decode", "gpp:polling", "gpp:gather",
                                     "gpp:stage8", "gpp:stage9", "gpp:stage10", "gpp:stage11", "gpp:stage12", "gpp:stage13", "gpp:stage14", "gpp:stage15"};

}  // namespace

extern "C" int gpp_plan_run(const gpp_plan_op* ops, int n_ops, void* stream, void* const* events, int n_events)
{
    if (!ops || n_ops < 0) return GPP_ERR_BAD_ARG;
    hipStream_t main_st = (hipStream_t)stream;
    int ev = 0;
    bool active[kLanes] = {false, false};
    bool uses_lanes = false;
    for (int i = 0; i < n_ops; ++i) uses_lanes = uses_lanes || ((ops[i].kind >> 8) & 0xff) != 0;
    int dev = 0;
    if (uses_lanes) {
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        if (dev < 0 || dev >= kMaxDevices) return GPP_ERR_UNSUPPORTED;
    }
    Lanes& g_lanes = g_lanes_of_device[dev];
    std::unique_lock<std::mutex> guard(g_lanes.lock, std::defer_lock);
    if (uses_lanes) guard.lock();
    // every way out of this function -- the end of the plan AND an error in the middle of it -- first makes the caller's stream wait for
    // the side lanes that are still open: a failed run must not leave forked work that nothing downstream is ordered after (and a
    // stream capture that is under way keeps a well-formed fork / join structure)
    auto close_lanes = [&]() -> int {
        int first = GPP_OK;
        for (int m = 0; m < kLanes; ++m) {
            if (!active[m]) continue;
            hipError_t e = hipEventRecord(g_lanes.done[m], g_lanes.stream[m]);
            if (e == hipSuccess) e = hipStreamWaitEvent(main_st, g_lanes.done[m], 0);
            if (e != hipSuccess && first == GPP_OK) first = (int)e;
            active[m] = false;
        }
        return first;
    };
    static const Roctx roctx;
    int stage_open = 0;
    auto set_stage = [&](int s) {
        if (!roctx.push || s == stage_open) return;
        if (roctx.sync) (void)hipDeviceSynchronize();
        if (stage_open) (void)roctx.pop();
        if (s) (void)roctx.push(kStageNames[s]);
        stage_open = s;
    };
    auto fail = [&](int rc) -> int { set_stage(0); (void)close_lanes(); return rc; };
    for (int i = 0; i < n_ops; ++i) {
        gpp_plan_op op = ops[i];
        if (!op.desc) return fail(GPP_ERR_BAD_ARG);
        set_stage((op.kind >> 20) & 15);
        op.kind &= 0xfffff;
        const int lane = (op.kind >> 8) & 0xff;
        const bool join = (op.kind & GPP_OP_JOIN) != 0, sync = (op.kind & GPP_OP_SYNC) != 0;
        op.kind &= 0xff;
        if (lane > kLanes) return fail(GPP_ERR_BAD_ARG);
        if (lane > 0 && join) return fail(GPP_ERR_BAD_ARG);      // only an op on the caller's stream can consume what the lanes produced
        hipStream_t st = main_st;
        if (lane > 0) {
            int rc = g_lanes.init();
            if (rc != GPP_OK) return fail(rc);
            st = g_lanes.stream[lane - 1];
            if (!active[lane - 1] || sync) {            // fork: the lane starts after everything enqueued on the main stream so far
                hipError_t e = hipEventRecord(g_lanes.fork, main_st);
                if (e == hipSuccess) e = hipStreamWaitEvent(st, g_lanes.fork, 0);
                if (e != hipSuccess) return fail((int)e);
                active[lane - 1] = true;
            }
        } else if (join) {                              // this op consumes what the side lanes produced
            int rc = close_lanes();
            if (rc != GPP_OK) return rc;
        }
        void* stream = (void*)st;
        const bool timed = op.tag != 0 && events && ev + 1 < n_events;
        if (timed) {
            hipError_t e = hipEventRecord((hipEvent_t)events[ev], st);
            if (e != hipSuccess) return fail((int)e);
        }
        int rc = GPP_ERR_UNSUPPORTED;
        switch (op.kind) {
        case GPP_OP_STEM: {
            const gpp_stem_desc* d = (const gpp_stem_desc*)op.desc;
            if (d->dtype == GPP_F32)      // reference-precision path: float32 [147][64] weights, fmaf chain on the vector ALUs
                rc = gpp_stem_conv7x7_bn_relu(d->in, (const float*)d->weight, d->bias, d->out, d->dtype, d->B, d->H, d->W, stream);
            else if (d->dtype == GPP_F16X3 || d->dtype == GPP_BF16X3)      // float32 output, three half products per float32 product
                rc = gpp_stem_conv7x7_bn_relu_x3_rc(d->in, d->weight, d->bias, (float*)d->out, d->B, d->H, d->W, d->range_counter, stream);
            else
                rc = gpp_stem_conv7x7_bn_relu_mfma(d->in, d->weight, d->bias, d->out, d->dtype, d->B, d->H, d->W, stream);
            break;
        }
        case GPP_OP_STEM_POOL: {
            const gpp_stem_desc* d = (const gpp_stem_desc*)op.desc;
            if (d->dtype == GPP_F16X3 || d->dtype == GPP_BF16X3)
                rc = gpp_stem_pool_fused_x3(d->in, d->weight, d->bias, (float*)d->out, d->B, d->H, d->W, d->range_counter, stream);
            else
                rc = gpp_stem_pool_fused_mfma(d->in, d->weight, d->bias, d->out, d->dtype, d->B, d->H, d->W, stream);
            break;
        }
        case GPP_OP_MAXPOOL: {
            const gpp_pool_desc* d = (const gpp_pool_desc*)op.desc;
            rc = gpp_maxpool3x3s2_same(d->in, d->out, d->dtype, d->B, d->H, d->W, d->C, stream);
            break;
        }
        case GPP_OP_CONV:
            rc = gpp_conv2d_igemm((const gpp_conv_desc*)op.desc, stream);
            break;
        case GPP_OP_BOTTLENECK_TAIL: {
            const gpp_tail_desc* d = (const gpp_tail_desc*)op.desc;
            rc = gpp_bottleneck_tail(d->conv3x3, d->conv1x1, d->tile_rows, stream);
            break;
        }
        case GPP_OP_BOTTLENECK_BLOCK: {
            const gpp_block_desc* d = (const gpp_block_desc*)op.desc;
            rc = gpp_bottleneck_block(d->conv1x1_a, d->conv3x3_b, d->conv1x1_c, d->tile, stream);
            break;
        }
        case GPP_OP_RELU: {
            const gpp_relu_desc* d = (const gpp_relu_desc*)op.desc;
            rc = gpp_relu_strided(d->in, d->in_bstride, d->out, d->out_bstride, d->dtype, d->B, d->count, stream);
            break;
        }
        case GPP_OP_DETECT: {
            const gpp_detect_desc* d = (const gpp_detect_desc*)op.desc;
            rc = gpp_detect_f32(d->cls_logits, d->regression, d->regression_dim, d->anchors, d->B, d->n_anchors,
                                d->num_base_anchors, d->fused_layout, d->score_thr, d->iou_thr, d->max_det, d->boxes, d->dims,
                                d->scores, d->labels, d->orientations, d->anchor_index, d->counts, d->workspace,
                                d->workspace_bytes, stream);
            break;
        }
        case GPP_OP_DETECT_OSF: {
            const gpp_detect_desc* d = (const gpp_detect_desc*)op.desc;
            rc = gpp_detect_osf_f32(d->cls_logits, d->regression, d->regression_dim, d->anchors, d->B, d->n_anchors,
                                    d->num_base_anchors, d->fused_layout, d->score_thr, d->iou_thr, d->max_det, d->boxes, d->dims,
                                    d->scores, d->labels, d->orientations, d->anchor_index, d->counts, d->workspace,
                                    d->workspace_bytes, stream);
            break;
        }
        case GPP_OP_DETECT_CANDIDATES:
        case GPP_OP_DETECT_SELECT:
        case GPP_OP_DETECT_EMIT: {
            const gpp_detect_desc* d = (const gpp_detect_desc*)op.desc;
            const int stages = op.kind == GPP_OP_DETECT_CANDIDATES ? GPP_DETECT_CANDIDATES
                             : op.kind == GPP_OP_DETECT_SELECT ? GPP_DETECT_SELECT : GPP_DETECT_EMIT;
            rc = gpp_detect_stages_f32(stages, d->cls_logits, d->regression, d->regression_dim, d->anchors, d->B, d->n_anchors,
                                       d->num_base_anchors, d->fused_layout, d->score_thr, d->iou_thr, d->max_det, d->boxes,
                                       d->dims, d->scores, d->labels, d->orientations, d->anchor_index, d->counts, d->workspace,
                                       d->workspace_bytes, stream);
            break;
        }
        case GPP_OP_POLL: {
            const gpp_poll_desc* d = (const gpp_poll_desc*)op.desc;
            rc = gpp_poll_f32(d->boxes, d->dims, d->orient, d->P_inv, d->planes, d->B, d->D, d->N, d->planes_batched, d->thr,
                              d->keypoints, d->keyplanes, d->residuals, d->best_idx, d->workspace, d->workspace_bytes, stream);
            break;
        }
        default:
            return fail(GPP_ERR_BAD_ARG);
        }
        if (rc != GPP_OK) return fail(rc);
        if (timed) {
            hipError_t e = hipEventRecord((hipEvent_t)events[ev + 1], st);
            if (e != hipSuccess) return fail((int)e);
            ev += 2;
        }
    }
    set_stage(0);
    return close_lanes();                               // a plan may not end inside a side lane
}

extern "C" int gpp_event_create(void** event)
{
    if (!event) return GPP_ERR_BAD_ARG;
    hipEvent_t e;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) return (int)rc;
    *event = (void*)e;
    return GPP_OK;
}

extern "C" int gpp_event_destroy(void* event)
{
    if (!event) return GPP_ERR_BAD_ARG;
    hipError_t rc = hipEventDestroy((hipEvent_t)event);
    return rc == hipSuccess ? GPP_OK : (int)rc;
}

extern "C" int gpp_event_elapsed_ms(void* start, void* stop, float* ms)
{
    if (!start || !stop || !ms) return GPP_ERR_BAD_ARG;
    hipError_t rc = hipEventSynchronize((hipEvent_t)stop);
    if (rc != hipSuccess) return (int)rc;
    rc = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    return rc == hipSuccess ? GPP_OK : (int)rc;
}
