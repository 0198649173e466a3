import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ['BXH_SWAVE'] = '1'
import numpy as np
from baryonyx_amd import _ffi
from oracle import oracle_ffi as O
import test_gpu_swave as T
inst = T.sat_like(700)
orc = O.Solver(inst, float_type=1, sort_mode=1, cost_norm=4, seed=1000)
dev = _ffi.DeviceBatch(inst, orc.cost(), 1, float_type=1)
print('form', dev.kernel_form)
from helpers import random_x
x0 = random_x(inst.n, 0.5, 77)
orc.set_x(x0); dev.set_x(0, x0); dev.set_rng(0, orc.rng); orc.order_init(0); dev.order_init(0)
vl = orc.violated().copy()
print('violated', len(vl), list(vl[:20]))
delta = orc.compute_delta(0.5)
# replay the oracle row by row, find the first row after which the device's final P differs in that row
st = dev.sweep(np.array([0.0]), delta, 0.5)
Pd = dev.get_P(0); pid = dev.get_pi(0)
for pos, k in enumerate(vl):
    orc.update_rows([int(k)], 0.0, delta, 0.5)
    e0, e1 = inst.row_ptr[k], inst.row_ptr[k+1]
    Po = orc.get_P()
    if not np.array_equal(Po[e0:e1].view(np.uint64), Pd[e0:e1].view(np.uint64)) or orc.get_pi()[k] != pid[k]:
        print('first diff at list pos', pos, 'row', k, 'cols', inst.col[e0:e1], 'fac', inst.factor[e0:e1], 'bounds', inst.cmin[k], inst.cmax[k])
        print(' oracle P', Po[e0:e1], 'pi', orc.get_pi()[k]); print(' device P', Pd[e0:e1], 'pi', pid[k])
        print(' neighbours in list', list(vl[max(0,pos-4):pos+3]))
        break
else:
    print('no row differs; rng', st['rng'][0], orc.rng)
