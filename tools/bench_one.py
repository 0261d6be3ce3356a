""" One conv shape, few launches (for rocprofv3 --pmc runs). usage: bench_one.py tile [Cin Cout K] """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench_conv as bc
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 512
k = int(sys.argv[4]) if len(sys.argv) > 4 else 3
bc.bench('conv %dx%d %d->%d tile %d' % (k, k, cin, cout, tile), 8, bc.PYR, cin, cout, k, tile=tile, iters=5)
