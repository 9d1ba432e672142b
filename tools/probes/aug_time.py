import sys, time; sys.path.insert(0, '.')
import torch
from keypoints_amd.data_augments import TpsAndRotate
aug = TpsAndRotate(4, 0.05, 0.1)
x = torch.rand(128, 3, 128, 128).cuda()
for _ in range(5): aug(x, x)
torch.cuda.synchronize(); t = time.time()
for _ in range(50): aug(x, x)
torch.cuda.synchronize(); print('augment ms per call', (time.time() - t) / 50 * 1e3)
