import sys, os, time
sys.path.insert(0, 'ground-plane-polling_amd')
import torch, numpy as np
from keras_retinanet_3D.backend import hip
B,H,W=8,402,1333
x=torch.randn((B,H,W,3),device='cuda')*50
w=hip.pack_stem_weights(np.random.default_rng(0).normal(size=(147,64)).astype(np.float32)*0.05, torch.device('cuda'))
b=torch.zeros((64,),device='cuda')
out=torch.empty((B,201,667,64),dtype=torch.bfloat16,device='cuda')
def run(): hip.check(hip.lib().gpp_stem_conv7x7_bn_relu_mfma(hip.ptr(x),hip.ptr(w),hip.ptr(b),hip.ptr(out),1,B,H,W,hip.stream_ptr()))
for _ in range(5): run()
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(os.environ.get('GPP_STEM_WGS_PER_CU','2'), 'WGs/CU: stem %.1f us'%(e0.elapsed_time(e1)*20))
pooled=torch.empty((B,101,334,64),dtype=torch.bfloat16,device='cuda')
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1000.0/n
def pool(): hip.check(hip.lib().gpp_maxpool3x3s2_same(hip.ptr(out),hip.ptr(pooled),1,B,201,667,64,hip.stream_ptr()))
def both(): run(); pool()
def fused(): hip.check(hip.lib().gpp_stem_pool_fused_mfma(hip.ptr(x),hip.ptr(w),hip.ptr(b),hip.ptr(pooled),1,B,H,W,hip.stream_ptr()))
print('pool alone %.1f us; stem + pool %.1f us; fused stem+pool %.1f us' % (timed(pool), timed(both), timed(fused)))

if hasattr(hip.lib(), 'gpp_debug_set_stem_stamps'):
    import ctypes
    st = torch.zeros((64 * 16, 8), dtype=torch.int64, device='cuda')
    hip.lib().gpp_debug_set_stem_stamps(ctypes.c_void_p(st.data_ptr()))
    run(); torch.cuda.synchronize()
    a = st.cpu().numpy().astype(np.float64) * 0.01
    a = a[a[:, 0] > 0][:, :6]
    d = np.diff(a, axis=1)
    nxt = a[1:, 0] - a[:-1, 5]
    print('per tile (us): wait for the previous tile readers %.2f  patch regs -> LDS %.2f  issue next patch loads %.2f  LDS reads + MFMA %.2f  stores %.2f ; tile total %.2f (%d tiles stamped)' %
          (d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 3].mean(), d[:, 4].mean(), (a[:, 5] - a[:, 0]).mean(), len(a)))
