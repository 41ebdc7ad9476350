import torch, sys
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0,'/root/repo')
import tests.test_kernels_gpu as tk
for cfg in [(2, 600, 24, 144, False, 0.3), (2, 600, 40, 240, False, 0.3), (2, 150, 64, 384, False, 0.3), (2, 40, 176, 1056, False, 0.2), (2, 600, 24, 144, True, 0.3)]:
    try:
        tk.test_bn_fold_expand_backward(*cfg)
        print("ok", cfg)
    except AssertionError as e:
        print("FAIL", cfg, str(e)[:200])
