"""driver.main alone in a fresh process on the files tools/driver_bench.py left in /tmp/vmx_driver_bench (run that first)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vacmap_amd import driver
d = '/tmp/vmx_driver_bench'
t = time.time()
driver.main(['-ref', d + '/ref.fa', '-read', d + '/' + (sys.argv[1] if len(sys.argv) > 1 else 'reads_quarter.fq'), '-mode', 'H', '-o', d + '/out_alone.sam', '-t', '16', '--nowriteindex', '--force'])
print('wall %.2f' % (time.time() - t))
