import os, sys, importlib.util, traceback
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,R)
import numpy as np
spec=importlib.util.spec_from_file_location('fz',os.path.join(R,'tests/test_gpu_fuzz.py')); m=importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
ch=np.load(os.path.join(R,"tests/golden/channels_vehA.npz"))["h"]
bad=0; n=0
seeds = range(1, 1 + int(sys.argv[1])) if len(sys.argv) > 1 else (1, 2, 3, 4)      # python tools/fuzz_big.py [n_seeds]: 100 geometries each
for seed in seeds:
    for case in m._cases(100, seed=seed*7919):
        n+=1
        try:
            m.test_random_geometry(ch, *case)
        except Exception as e:
            bad+=1; print("FAIL", case, repr(e)[:300]); 
print("cases", n, "failures", bad)
