import os, sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
buf = torch.zeros(3 * 64, dtype=torch.int64, device='cuda:0')
os.environ['STCD_RES_STAMPS'] = str(buf.data_ptr())
sys.argv = ['opbench.py'] + sys.argv[1:]
os.environ['OPBENCH_KIND'] = 'conv'
import runpy
runpy.run_path('/root/repo/tools/opbench.py', run_name='__main__')
torch.cuda.synchronize()
b = buf.cpu().numpy().reshape(3, 64)
for k in range(3):
    row = b[k][b[k] > 0]
    d = np.diff(row)
    print('block', k, 'total', int(row[-1] - row[0]), 'deltas', [int(x) for x in d])
