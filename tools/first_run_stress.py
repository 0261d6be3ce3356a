""" The scenario in which the wrong-plane transient of round 2 was seen, repeated: fresh processes sharing the one GPU, each building
its own model and computing its 2-image shard ONCE (tests/test_zz_sharded_gpu.py), with best_index checked against the CPU oracle
on the run's own boxes in every process, and -- with the diagnostic library (make -C ground-plane-polling_amd/csrc polldbg,
GPP_LIB=.../libgpp_hip_polldbg.so) -- the kernel's own record of what each wavefront saw.

    python tools/first_run_stress.py drive <rounds> <procs per round> [dtype]
Environment: GPP_TUNE_CACHE (skips the timing runs of the tile choice: a round then takes ~10 s instead of ~60 s),
STRESS_NO_DIST=1 (no gloo group), STRESS_OUT (directory for the dumps of failing runs).
"""
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'ground-plane-polling_amd'), os.path.join(ROOT, 'tests'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def worker(rank, world, port, dtype, tag):
    import ctypes
    import numpy as np
    import torch
    import helpers
    import sharded_worker
    torch.cuda.set_device(0)
    use_dist = os.environ.get('STRESS_NO_DIST', '0') == '0'
    if use_dist:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank))
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from keras_retinanet_3D import models
    from keras_retinanet_3D.backend import hip
    from keras_retinanet_3D.utils import distributed as D
    batch, h, w = 2 * world, 402, 1333
    model = models.load_model('synthetic:1234', backbone_name='resnet50', dtype=dtype)
    inputs = list(sharded_worker.global_inputs(batch, h, w))
    lo, hi = D.shard_range(batch, rank, world)
    dbg = None
    if hasattr(hip.lib(), 'gpp_poll_debug_buffer'):
        dbg = torch.full(((hi - lo) * 100, 256, 8), float('nan'), dtype=torch.float32, device='cuda')      # per lane: csrc/poll.hip, GPP_POLL_DEBUG
        hip.lib().gpp_poll_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    if use_dist:
        D.ShardedModel(model).predict_on_batch(inputs)
    else:
        model.predict_on_batch([a[lo:hi] for a in inputs])
    plan = model.plan_for(hi - lo, h, w, 1000, True)
    torch.cuda.synchronize()
    boxes, dims, orient = plan.boxes.cpu().numpy(), plan.dimensions.cpu().numpy(), plan.orientations.cpu().numpy()
    best, kp, res = plan.best_index.cpu().numpy(), plan.keypoints.cpu().numpy(), plan.residuals.cpu().numpy()
    oracle = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'liboracle_polling.so'))
    want = helpers.c_oracle_poll(oracle, boxes, dims, orient, inputs[1][lo:hi], inputs[2][lo:hi])
    ok = np.array_equal(best, want[3]) and helpers.bits_equal(kp, want[0]) and helpers.bits_equal(res, want[2])
    import hashlib
    digest = hashlib.sha1(b''.join(np.ascontiguousarray(t.cpu().numpy()).tobytes() for t in model.outputs(plan))).hexdigest()[:16]
    print('OUTPUT_SHA rank {} {}'.format(rank, digest), flush=True)
    if not ok:
        out = os.environ.get('STRESS_OUT', os.path.join(ROOT, 'gpurun_out', 'r3'))
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, 'firstrun_fail_{}_rank{}.npz'.format(tag, rank))
        np.savez(path, boxes=boxes, dims=dims, orient=orient, best=best, kp=kp, res=res, want_best=want[3], want_kp=want[0], want_res=want[2],
                 canon=plan.poll_ws.cpu().numpy().view(np.float32), planes=plan.planes.cpu().numpy(), P_inv=plan.P_inv.cpu().numpy(),
                 dbg=dbg.cpu().numpy() if dbg is not None else np.zeros(0, np.float32))
        for b, d in np.argwhere(best != want[3])[:6]:
            line = 'WRONG PLANE tag {} rank {} image {} detection {}: gpu {} (residual {:.6f}) oracle {} (residual {:.6f})'.format(
                tag, rank, lo + b, d, best[b, d], res[b, d] * 6, want[3][b, d], want[2][b, d] * 6)
            if dbg is not None:
                q = dbg[b * 100 + d].cpu().numpy()                                   # (256 lanes, 8): rmin, imin, i100, level, exec lo / hi, iterations, tid
                act = q[:, 4:6].view(np.uint32)
                line += '\n    lane states: active masks {}, iterations {}, lanes whose level is below the maximum {}'.format(
                    sorted(set('%08x%08x' % (hi_, lo_) for lo_, hi_ in act.tolist())), sorted(set(q[:, 6].view(np.int32).tolist())),
                    int((q[:, 3].view(np.int32) < q[:, 3].view(np.int32).max()).sum()))
            print(line, flush=True)
        print('dumped ' + path, flush=True)
    else:
        print('tag {} rank {}: ok ({} detections above 0.05)'.format(tag, rank, int((plan.scores.cpu().numpy() > 0.05).sum())), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def drive(rounds, procs, dtype):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    bad = 0
    t0 = time.time()
    shas = {}
    for r in range(rounds):
        if os.environ.get('STRESS_RANDOM_TILES'):
            env['GPP_TUNE_RANDOM'] = str(int(os.environ['STRESS_RANDOM_TILES']) + r)
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), 'worker', str(k), str(procs), str(port), dtype, 'r%d' % r], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True) for k in range(procs)]
        for p in ps:
            try:
                out = p.communicate(timeout=600)[0]
            except subprocess.TimeoutExpired:
                for q in ps:
                    q.kill()
                raise
            lines = [l for l in out.splitlines() if 'amdgpu.ids' not in l and l.strip()]
            for l in lines:
                if l.startswith('OUTPUT_SHA'):
                    shas.setdefault(l.split()[2], set()).add(l.split()[3])
            print('\n'.join(lines[-12:]), flush=True)
            if p.returncode != 0:
                bad += 1
        print('--- round {} done, {} failing processes so far, {:.0f} s'.format(r, bad, time.time() - t0), flush=True)
    print('first_run_stress: {} rounds x {} processes, {} failing processes; distinct output hashes per rank: {}'.format(
        rounds, procs, bad, {k: sorted(v) for k, v in sorted(shas.items())}), flush=True)
    return 1 if bad or any(len(v) != 1 for v in shas.values()) else 0


if __name__ == '__main__':
    if sys.argv[1] == 'drive':
        sys.exit(drive(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else 'bf16'))
    sys.exit(worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]))
