import os, sys
print({k:v for k,v in os.environ.items() if 'VISIBLE' in k or 'HSA' in k or 'HIP' in k or 'ROC' in k})
order = sys.argv[1]
sys.path.insert(0, '.')
import importlib
if order == 'lib_first':
    pkg = importlib.import_module("gpu-icp-slam_amd"); print('lib devcount', pkg.device_count())
    import torch; print('torch', torch.cuda.device_count(), torch.cuda.is_available())
else:
    import torch; print('torch', torch.cuda.device_count(), torch.cuda.is_available())
    pkg = importlib.import_module("gpu-icp-slam_amd"); print('lib devcount', pkg.device_count())
