import sys, torch, torch.nn.functional as Fn
sys.path.insert(0, '/root/repo')
from tf_eager_object_detection_amd.model import fpn_detector as fd
from tf_eager_object_detection_amd.model.frcnn_detector import ResNetC4Detector, Vgg16Detector
fd._CONV3X3_MODE = 'force'
for fam in ('c4', 'vgg16'):
    torch.manual_seed(2)
    shape = (320, 480)
    m = (ResNetC4Detector(50, 21, shape, 64, dtype=torch.float32, max_batch=2) if fam == 'c4' else Vgg16Detector(21, shape, 64, dtype=torch.float32, max_batch=2)).prepare()
    img = torch.randn((2,) + shape + (3,), device='cuda') * 50
    m(img)
    calls = []
    saved = {}
    for name in ('conv2d', 'linear'):
        real = getattr(Fn, name); saved[(Fn, name)] = real
        setattr(Fn, name, lambda *a, _r=real, _n=name, **k: calls.append((_n, tuple(a[0].shape), tuple(a[1].shape))) or _r(*a, **k))
    for name in ('addmm', '_addmm_activation', 'matmul', 'mm'):
        real = getattr(torch, name); saved[(torch, name)] = real
        setattr(torch, name, lambda *a, _r=real, _n=name, **k: calls.append((_n,)) or _r(*a, **k))
    m(img)
    for (mod, name), real in saved.items(): setattr(mod, name, real)
    print(fam, 'fp32 library calls:', calls)
