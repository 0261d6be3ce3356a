""" Aggregate tools/pmc_bench.sh output for the dominant kernel -> profiles/<round>/dominant_kernel_pmc_<dtype>.{txt,json}
    python tools/pmc_aggregate.py <dir with the counter runs> <output prefix> [dtype] """
import collections
import csv
import glob
import json
import sys

src, out_prefix = sys.argv[1], sys.argv[2]
dtype = sys.argv[3] if len(sys.argv) > 3 else 'f16x3'
CODE = {'bf16': 1, 'f16': 2, 'f32': 3, 'bf16x3': 4, 'f16x3': 5}[dtype]
ESZ = 2 if dtype in ('bf16', 'f16') else 4
# the regression-tower launches: the uniform 256 x 256 grid (722 workgroups) or, where the autotuner picked it, the mixed-height grid of
# round 4 (conv_igemm_mix_kernel<dt, 256, 224, ...>: 512 + 236 workgroups); same layer, same bytes, same FLOPs
KERNELS = {'conv_igemm_kernel<%d, 256, 256' % CODE: 722 * 512, 'conv_igemm_mix_kernel<%d, 256, 224' % CODE: 748 * 512}
# the library build the counters were collected with (bench.py prints it in roofline.library): bench.py only reports the
# traffic figure when this matches the running library
version = None
for f in glob.glob(src + '/*.log'):
    for line in open(f):
        if line.startswith('{"metric"'):
            version = json.loads(line)['roofline'].get('library', version)
GRID = 722 * 512
agg = collections.defaultdict(list)
forms = collections.Counter()
for f in glob.glob(src + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        for kernel, grid in KERNELS.items():
            if kernel in r['Kernel_Name'] and int(r['Grid_Size']) == grid:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
                forms[kernel.split('<')[0]] += 1
mean = {k: sum(v) / len(v) for k, v in agg.items()}
read_b = 2.0 * mean['FETCH_SIZE'] * 1024
write_b = mean['WRITE_SIZE'] * 1024
alg_read, alg_write = 91504 * 512 * ESZ + 512 * 4608 * ESZ, 91504 * 512 * ESZ
lines = ['library: %s' % version, 'dominant kernel = conv_igemm_kernel<%s,256,256,2,4,2,pipe%s>, grid 722 x 512 threads' % (dtype, ',pre-split input' if ESZ == 4 else ''),
         '(regression tower 3x3 512->512 over the 5 pyramid levels, M = 91504 rows, B = 8); launches by grid form: %s' % dict(forms),
         'collected with tools/pmc_bench.sh: rocprofv3 --kernel-trace --pmc <one group per run> -- python3 bench.py --dtype %s --steps 4 --warmup 2 --no-cpu-baseline' % dtype, '']
for k in sorted(mean):
    lines.append('%-28s launches=%3d  mean per launch %.6g' % (k, len(agg[k]), mean[k]))
lines += ['', 'Fabric-side traffic per launch (MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE count KiB at the L2 <-> fabric',
          'interface, Infinity-Cache hits included; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide 16 B/lane streams -> doubled;',
          'WRITE_SIZE is exact for 16 B/lane stores):',
          '  read  = 2 * FETCH_SIZE * 1024 = %.1f MB   (algorithmic %.1f MB = activations once + weights once)' % (read_b / 1e6, alg_read / 1e6),
          '  write = WRITE_SIZE * 1024     = %.1f MB   (algorithmic %.1f MB)' % (write_b / 1e6, alg_write / 1e6),
          '  total = %.1f MB' % ((read_b + write_b) / 1e6)]
if 'TCC_HIT_sum' in mean:
    lines.append('L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS) = %.3f' % (mean['TCC_HIT_sum'] / (mean['TCC_HIT_sum'] + mean['TCC_MISS_sum'])))
if 'SQ_VALU_MFMA_BUSY_CYCLES' in mean and 'GRBM_GUI_ACTIVE' in mean:
    lines.append('MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) = %.3f' % (mean['SQ_VALU_MFMA_BUSY_CYCLES'] / (mean['GRBM_GUI_ACTIVE'] / 8 * 1024)))
if 'SQ_INSTS_VALU' in mean:
    # SQ_INSTS_VALU counts the matrix instructions too (they are VALU-class: on the register-only MFMA loop of tools/micro/mfma_power.hip
    # SQ_INSTS_VALU / SQ_INSTS_MFMA = 1.0x, profiles/r4/valu_counter_includes_mfma.txt); the vector-ALU work BESIDE the matrix pipe is the difference
    lines.append('SQ_INSTS_VALU / SQ_INSTS_MFMA = %.2f (the counter includes the MFMAs) -> non-MFMA VALU instructions per MFMA = %.2f' % (
        mean['SQ_INSTS_VALU'] / mean['SQ_INSTS_MFMA'], mean['SQ_INSTS_VALU'] / mean['SQ_INSTS_MFMA'] - 1.0))
    lines.append('(disassembly of the K-step of the 256 x 256 tile: 96 MFMA, 24 ds_read_b128, 8 LDS-DMA, 24 other VALU -- tap masks and LDS addresses -- per wavefront)')
open(out_prefix + '.txt', 'w').write('\n'.join(lines) + '\n')
json.dump({'kernel': 'conv_igemm_kernel<%s,256,256,2,4,2,pipe>' % dtype, 'dtype': dtype, 'grid': GRID, 'library_version': version, 'traffic_bytes_per_launch': read_b + write_b,
           'read_bytes': read_b, 'write_bytes': write_b, 'algorithmic_bytes': alg_read + alg_write,
           'counters': mean}, open(out_prefix + '.json', 'w'), indent=1)
print('\n'.join(lines))
