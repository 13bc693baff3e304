import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import test_gpu_reference_tables as t
res = []
for s in range(1, 11):
    res.append(t._run("gmm", 8, s))
r = np.array(res)
print("GMM8 elbo", np.round(r[:, 0], 3).tolist(), "lnZ", np.round(r[:, 1], 3).tolist(), "mean", r.mean(0), "std", r.std(0, ddof=1))
