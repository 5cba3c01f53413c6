"""Persistent per-tile kernel (ICNN_BE_FLAG_PERSISTENT) against one launch per phase and round
(ICNN_BE_FLAG_TWO_KERNELS) over batch sizes, Bibsonomy shape, nIter = 10 (GPU box only): the data behind the
automatic choice in icnn_be_solve_fc."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icnn_amd import _lib, bundle_entropy, picnn
spec=picnn.bibtex_spec(); params=picnn.init_params(spec,0,'spread')
for B in (16384, 8192, 4096, 2048, 1024, 512, 128):
    x=torch.from_numpy((np.random.RandomState(1000).rand(B,spec.n_features)<0.04).astype(np.float32)).cuda()
    model=picnn.FCModel(spec,params); ctx=model.context(x)
    out={}
    for name,flags in (('fused',_lib.FLAG_PERSISTENT),('two',_lib.FLAG_TWO_KERNELS)):
        sol=bundle_entropy.FusedSolver(model,B,10,'dual',flags=flags)
        for _ in range(3): res=sol.solve(ctx,0.5)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(10): res=sol.solve(ctx,0.5)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
        out[name]=(res.y.cpu().numpy().copy(), res.count[:B].cpu().numpy().copy(), res.newton_iters[:B].cpu().numpy().copy(), res.lam.cpu().numpy().copy(), dt)
    a,b=out['fused'],out['two']
    print('B=%d fused %.3f ms  two-kernel %.3f ms  y equal %s  count equal %s newton equal %s lam equal %s max|dy| %.2e'%(B,a[4]*1e3,b[4]*1e3,np.array_equal(a[0],b[0]),np.array_equal(a[1],b[1]),np.array_equal(a[2],b[2]),np.array_equal(a[3],b[3]),np.abs(a[0]-b[0]).max()))
