"""one layer shape of odet_conv3x3_f16 for --pmc passes: python tools/exp/conv3x3_one.py [H W cin cout B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd import ops
a = [int(v) for v in sys.argv[1:]] + [200, 334, 256, 512, 8][len(sys.argv) - 1:]
H, W, cin, cout, B = a[:5]
x = torch.randn(B, H, W, cin, device='cuda').half()
w = (torch.randn(cout, cin, 3, 3, device='cuda') * 0.01).half().contiguous(memory_format=torch.channels_last)
for _ in range(12):
    y = ops.conv3x3_f16(x, w)
torch.cuda.synchronize()
