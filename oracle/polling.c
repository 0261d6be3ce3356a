/*
 * ORACLE (test infrastructure, not product code): plain-C restatement of the reference's
 * ground-plane polling, float32, one operation at a time (build with -ffp-contract=off).
 *
 * Follows /root/reference/keras_retinanet_3D/layers/fit_road_planes.py
 *     poll            :18-32
 *     calc_X_t        :34-47
 *     fit_road_planes :49-139
 * and must agree bit for bit with oracle/polling_np.py, which is pinned by the golden
 * vectors in tests/golden/polling_*.npz (reference executed on a NumPy stand-in).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 * Literal two-pass form: all per-plane votes / residuals / z-checks are materialised, then
 * masked and arg-min'ed exactly in the reference's order (:112-119).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <float.h>

typedef struct { float x, y, z; } v3;

static inline v3 sub3(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline v3 scale3(v3 a, float s) { v3 r = { a.x * s, a.y * s, a.z * s }; return r; }
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b)
{
    v3 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
    return r;
}
static inline float norm3(v3 a) { return sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z); }
static inline float sgn(float v) { return (float)((v > 0.0f) - (v < 0.0f)); }

/* fit_road_planes.py:75-77 */
void gpp_oracle_canonical_planes(const float *planes, int64_t n, float *out)
{
    for (int64_t j = 0; j < n; ++j) {
        float dir = -sgn(planes[4 * j + 1]);
        float a = planes[4 * j + 0] * dir, b = planes[4 * j + 1] * dir;
        float c = planes[4 * j + 2] * dir, d = planes[4 * j + 3] * dir;
        float nn = sqrtf((a * a + b * b) + c * c);
        out[4 * j + 0] = a / nn; out[4 * j + 1] = b / nn;
        out[4 * j + 2] = c / nn; out[4 * j + 3] = d / nn;
    }
}

typedef struct { v3 X[4]; float zc; float votes; float res; } hyp;

/* fit_road_planes.py:84-109 for one (detection, plane) pair */
static hyp evaluate(const v3 ray[4], const float *pl, const float target[6], float thr)
{
    hyp h;
    v3 n = { pl[0], pl[1], pl[2] };
    float d = pl[3];
    for (int k = 0; k < 3; ++k)
        h.X[k] = scale3(ray[k], fabsf((-d) / dot3(n, ray[k])));
    h.zc = cross3(sub3(h.X[0], h.X[1]), sub3(h.X[2], h.X[1])).y;
    v3 perp = cross3(ray[3], cross3(n, ray[3]));
    float num = dot3(perp, h.X[1]);
    float den = dot3(perp, n);
    h.X[3] = sub3(h.X[1], scale3(n, num / den));
    static const int seg[6][2] = { {1, 3}, {0, 1}, {1, 2}, {0, 2}, {0, 3}, {2, 3} };
    h.votes = 0.0f;
    h.res = 0.0f;
    for (int p = 0; p < 6; ++p) {
        float r = fabsf(norm3(sub3(h.X[seg[p][0]], h.X[seg[p][1]])) - target[p]);
        float v = (r > thr) ? 0.0f : 1.0f;
        h.votes = (p == 0) ? v : h.votes + v;
        h.res = (p == 0) ? r : h.res + r;
    }
    return h;
}

/*
 * boxes (B,D,12) dims (B,D,3) orient (B,D) P_inv (B,4,3) planes (N,4) or (B,N,4)
 * -> keypoints (B,D,4,3) keyplanes (B,D,4) residuals (B,D) best_idx (B,D)
 * returns 0, or -1 on bad arguments / allocation failure.
 */
int gpp_oracle_poll_f32(const float *boxes, const float *dims, const int32_t *orient, const float *P_inv,
                        const float *planes, int B, int D, int N, int planes_batched, float thr,
                        float *keypoints, float *keyplanes, float *residuals, int32_t *best_idx)
{
    if (B < 0 || D < 0 || N <= 0) return -1;
    float *canon = (float *)malloc(sizeof(float) * 4 * (size_t)N);
    float *R = (float *)malloc(sizeof(float) * (size_t)N);
    float *V = (float *)malloc(sizeof(float) * (size_t)N);
    float *Z = (float *)malloc(sizeof(float) * (size_t)N);
    if (!canon || !R || !V || !Z) { free(canon); free(R); free(V); free(Z); return -1; }
    for (int b = 0; b < B; ++b) {
        if (b == 0 || planes_batched)
            gpp_oracle_canonical_planes(planes + (planes_batched ? (size_t)b * N * 4 : 0), N, canon);
        const float *Pi = P_inv + (size_t)b * 12;
        for (int i = 0; i < D; ++i) {
            const float *bx = boxes + ((size_t)b * D + i) * 12;
            const float *dm = dims + ((size_t)b * D + i) * 3;
            int o = orient[(size_t)b * D + i];
            /* :80-83 back-projection, homogeneous component dropped, flipped to z > 0 */
            v3 ray[4];
            for (int k = 0; k < 4; ++k) {
                float x = bx[4 + 2 * k], y = bx[5 + 2 * k];
                float r0 = (Pi[0] * x + Pi[1] * y) + Pi[2] * 1.0f;
                float r1 = (Pi[3] * x + Pi[4] * y) + Pi[5] * 1.0f;
                float r2 = (Pi[6] * x + Pi[7] * y) + Pi[8] * 1.0f;
                float s = sgn(r2);
                ray[k].x = r0 * s; ray[k].y = r1 * s; ray[k].z = r2 * s;
            }
            /* :61-73,95-109 poll targets */
            float h = dm[0], w = dm[1], l = dm[2];
            float hw = sqrtf(h * h + w * w), wl = sqrtf(w * w + l * l), hl = sqrtf(h * h + l * l);
            float oh[4];
            for (int k = 0; k < 4; ++k) oh[k] = (o == k) ? 1.0f : 0.0f;
#define MIX(a, b, c, d) (((oh[0] * (a) + oh[1] * (b)) + oh[2] * (c)) + oh[3] * (d))
            float target[6] = { h, MIX(l, w, w, l), MIX(w, l, l, w), wl, MIX(hl, hw, hw, hl), MIX(hw, hl, hl, hw) };
#undef MIX
            float vmax = -1.0f;
            for (int j = 0; j < N; ++j) {
                hyp hy = evaluate(ray, canon + 4 * (size_t)j, target, thr);
                R[j] = hy.res; V[j] = hy.votes; Z[j] = hy.zc;
                if (hy.votes > vmax) vmax = hy.votes;
            }
            /* :112-119 */
            int best = 0;
            float bestv = FLT_MAX;
            for (int j = 0; j < N; ++j) {
                float r = R[j];
                if (V[j] - vmax < 0.0f) r = 100.0f;
                if (Z[j] < 0.0f) r = 100.0f;
                R[j] = r;
                if (r < bestv) { bestv = r; best = j; }   /* first minimum, NaN never wins */
            }
            hyp hy = evaluate(ray, canon + 4 * (size_t)best, target, thr);
            size_t row = (size_t)b * D + i;
            for (int k = 0; k < 4; ++k) {
                keypoints[row * 12 + 3 * k + 0] = hy.X[k].x;
                keypoints[row * 12 + 3 * k + 1] = hy.X[k].y;
                keypoints[row * 12 + 3 * k + 2] = hy.X[k].z;
            }
            for (int k = 0; k < 4; ++k) keyplanes[row * 4 + k] = canon[4 * (size_t)best + k];
            residuals[row] = R[best] / 6.0f;
            if (best_idx) best_idx[row] = best;
        }
    }
    free(canon); free(R); free(V); free(Z);
    return 0;
}
