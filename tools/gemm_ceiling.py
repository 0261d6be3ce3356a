""" Calibration only: what the vendor GEMM (hipBLASLt through torch.matmul) sustains on this GPU for the GEMM the
dominant conv is equivalent to, on random data, back to back -- the practical ceiling next to the 2.5 PFLOP/s peak. """
import torch

dev = torch.device('cuda')
for name, M, N, K in (('reg tower as GEMM', 91504, 512, 4608), ('towers_0 as GEMM', 91504, 896, 4608), ('cls tower as GEMM', 91504, 256, 2304),
                      ('square 8192', 8192, 8192, 8192)):
    a = (torch.randn((M, K), device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16)
    for layout, bb in (('B = W^T (weights [N][K], as the conv stores them)', b.t()), ('B row-major [K][N]', b.t().contiguous())):
        for _ in range(5):
            c = a @ bb
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            c = a @ bb
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        print('%-22s M %6d N %5d K %5d  %-52s %.3f ms  %7.1f TFLOP/s' % (name, M, N, K, layout, ms, 2.0 * M * N * K / ms / 1e9))
