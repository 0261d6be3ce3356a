// Sanitizer driver for the HOST side of libgpp_hip (make asan; tests/test_host_fuzz.py).  Test infrastructure, not part of the library.
//
// Every C-ABI entry point validates its descriptors on the host before anything reaches the device; the device-free ones -- gpp_conv2d_flops,
// gpp_conv2d_split_rule, gpp_conv2d_workspace_bytes, gpp_conv2d_tile_candidates, gpp_stem_pack_weights_f16 / _f16x3 -- and the argument checks of
// the launching ones (gpp_conv2d_igemm, gpp_bottleneck_tail, gpp_bottleneck_block, gpp_plan_run, gpp_poll_f32, ...: without a device they end
// in an error code before any launch) are run here over a file of descriptors the test generated, in a build of the library's host code
// with -fsanitize=address,undefined.  A bad descriptor must come back as GPP_ERR_* (or a hipError_t from the absent device); nothing may
// trip a sanitizer.  Device pointers inside the descriptors are never dereferenced by host code: they are fuzzed like every other field.
//
// File format: repeated records  [uint32 kind][uint32 a][uint32 b][uint32 c][gpp_conv_desc x 3]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "gpp.h"

static long g_rc_hist[3] = {0, 0, 0};      // GPP_OK, GPP_ERR_*, hipError_t

static void note(int rc)
{
    if (rc == 0) ++g_rc_hist[0];
    else if (rc < 0 && rc >= -4) ++g_rc_hist[1];
    else if (rc > 0) ++g_rc_hist[2];
    else { fprintf(stderr, "unexpected return code %d\n", rc); exit(3); }
}

struct Rec { uint32_t kind, a, b, c; gpp_conv_desc d[3]; };

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s records.bin  (record = %zu bytes, gpp_conv_desc = %zu)\n", argv[0], sizeof(Rec), sizeof(gpp_conv_desc)); return 2; }
    if (!strcmp(argv[1], "--sizes")) { printf("%zu %zu\n", sizeof(Rec), sizeof(gpp_conv_desc)); return 0; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    Rec r;
    long n = 0;
    while (fread(&r, sizeof r, 1, f) == 1) {
        ++n;
        double flops = 0;
        int split = 0, count = 0;
        size_t bytes = 0;
        note(gpp_conv2d_flops(&r.d[0], &flops));
        note(gpp_conv2d_split_rule(&r.d[0], &split));
        note(gpp_conv2d_workspace_bytes(&r.d[0], &bytes));
        {
            const int cap = (int)(r.a % 80);
            std::vector<int> tiles((size_t)cap + 1);      // exactly `cap` usable entries + a canary the library must not touch
            tiles[cap] = 0x5a5a5a5a;
            note(gpp_conv2d_tile_candidates(&r.d[0], tiles.data(), cap, &count));
            if (tiles[cap] != 0x5a5a5a5a) { fprintf(stderr, "tile_candidates wrote past its capacity\n"); return 3; }
        }
        note(gpp_conv2d_flops(nullptr, &flops));
        note(gpp_conv2d_tile_candidates(&r.d[0], nullptr, 0, &count));
        switch (r.kind % 6) {
        case 0: note(gpp_conv2d_igemm(&r.d[0], nullptr)); break;
        case 1: note(gpp_bottleneck_tail(&r.d[0], &r.d[1], (int)(r.b % 200), nullptr)); break;
        case 2: note(gpp_bottleneck_block(&r.d[0], &r.d[1], &r.d[2], (int)(r.b % 2000), nullptr)); break;
        case 3: {
            // a plan of a few ops over these descriptors (kinds and lane / join / sync flags fuzzed)
            gpp_tail_desc t = {&r.d[0], &r.d[1], (int32_t)(r.b % 200), 0};
            gpp_block_desc bl = {&r.d[0], &r.d[1], &r.d[2], (int32_t)(r.c % 2000), 0};
            // (a kind whose descriptor type the record does not carry gets a conv descriptor's bytes, or NULL: every entry point checks its own
            // arguments, and none of the other descriptor types is larger than a gpp_conv_desc)
            static_assert(sizeof(gpp_conv_desc) >= sizeof(gpp_detect_desc) && sizeof(gpp_conv_desc) >= sizeof(gpp_poll_desc) &&
                          sizeof(gpp_conv_desc) >= sizeof(gpp_stem_desc) && sizeof(gpp_conv_desc) >= sizeof(gpp_relu_desc) &&
                          sizeof(gpp_conv_desc) >= sizeof(gpp_pool_desc), "the stand-in bytes cover every descriptor type");
            gpp_plan_op ops[4];
            for (int i = 0; i < 4; ++i) {
                const uint32_t k = (r.a >> (8 * i)) & 0xff;
                ops[i].kind = (int32_t)((k % 3 == 0) ? GPP_OP_CONV : (k % 3 == 1) ? GPP_OP_BOTTLENECK_TAIL : GPP_OP_BOTTLENECK_BLOCK);
                if (((r.c >> (3 * i)) & 7) == 7) ops[i].kind = (int32_t)((r.c >> 3) & 0x1f);      // now and then: any kind at all
                const int kk = ops[i].kind & 0xff;
                if (k & 0x40) ops[i].kind |= GPP_OP_LANE((k >> 4) & 3);
                if (k & 0x80) ops[i].kind |= GPP_OP_JOIN;
                if ((r.b >> i) & 1) ops[i].kind |= GPP_OP_SYNC;
                ops[i].tag = (int32_t)(k & 1);
                ops[i].desc = kk == GPP_OP_CONV ? (const void*)&r.d[i % 3]
                            : kk == GPP_OP_BOTTLENECK_TAIL ? (const void*)&t
                            : kk == GPP_OP_BOTTLENECK_BLOCK ? (const void*)&bl : (((r.b >> (4 + i)) & 1) ? (const void*)&r.d[0] : nullptr);
            }
            note(gpp_plan_run(ops, (int)(r.b % 5), nullptr, nullptr, 0));
            note(gpp_plan_run(nullptr, 1, nullptr, nullptr, 0));
            note(gpp_plan_run(ops, -1, nullptr, nullptr, 0));
            break;
        }
        case 4: {
            // host-side weight packers: source [147][64] float32, destination of a fuzzed size
            std::vector<float> src(147 * 64);
            for (size_t i = 0; i < src.size(); ++i) src[i] = (float)((int)((r.a + 2654435761u * i) % 2001) - 1000) * ((r.b & 1) ? 1e-3f : 1e30f);
            const size_t need16 = 64 * 232 * 2, need_x3 = 2 * 64 * 232 * 2 + 64 * 4;
            const size_t sz = (r.c % 3 == 0) ? need16 : (r.c % 3 == 1) ? need_x3 : (size_t)(r.c % 70000);
            std::vector<unsigned char> dst(sz ? sz : 1);
            note(gpp_stem_pack_weights_f16(src.data(), dst.data(), sz));
            note(gpp_stem_pack_weights_f16x3(src.data(), dst.data(), sz));
            note(gpp_stem_pack_weights_f16(nullptr, dst.data(), sz));
            break;
        }
        default: {
            size_t wb = 0;
            note(gpp_poll_workspace_bytes((int)r.a, (int)r.b, (int)(r.c & 1), &wb));
            note(gpp_detect_workspace_bytes((int)r.a, (int64_t)(((uint64_t)r.b * (uint64_t)r.c) >> ((r.a >> 8) & 31)), &wb));
            note(gpp_detect_osf_workspace_bytes((int)r.a, (int64_t)(((uint64_t)r.b * (uint64_t)r.c) >> ((r.a >> 8) & 31)), &wb));
            note(gpp_poll_f32(nullptr, nullptr, nullptr, nullptr, nullptr, (int)r.a, (int)r.b, (int)r.c, 0, 0.7f, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr));
            break;
        }
        }
    }
    fclose(f);
    printf("%ld records: %ld GPP_OK, %ld GPP_ERR_*, %ld hipError_t\n", n, g_rc_hist[0], g_rc_hist[1], g_rc_hist[2]);
    return 0;
}
