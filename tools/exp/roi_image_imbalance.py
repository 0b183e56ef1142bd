import sys, json, numpy as np, torch
sys.path.insert(0,'/root/repo')
import bench
from tf_eager_object_detection_amd import synthetic as syn
from tf_eager_object_detection_amd.pipeline import FpnStepBatch, synthetic_fpn_inputs
shape=(800,1333)
host, dev = synthetic_fpn_inputs(shape, 21, 1000, 256, seed=1234)
B=8
sb = FpnStepBatch(B, shape, 21, 1000, 256)
gen = torch.Generator(device='cuda'); gen.manual_seed(4321)
for b in range(B):
    if b==0: d=dev
    else:
        pa = torch.randperm(dev['rpn_logits'].shape[0], device='cuda', generator=gen)
        pr = torch.randperm(dev['cls_scores'].shape[0], device='cuda', generator=gen)
        d = dict(rpn_logits=dev['rpn_logits'][pa].contiguous(), rpn_deltas=dev['rpn_deltas'][pa].contiguous(),
                 feats=[torch.randn(f.shape, device='cuda', dtype=torch.float32, generator=gen) for f in dev['feats']],
                 cls_scores=dev['cls_scores'][pr].contiguous(), cls_deltas=dev['cls_deltas'][pr].contiguous())
    sb.bind(b, d['rpn_logits'], d['rpn_deltas'], d['feats'], d['cls_scores'], d['cls_deltas'])
sb.enqueue(1, B); torch.cuda.synchronize()
rows=[]
for h in sb.slots:
    k=int(h.roi_count.item())
    a=bench.algorithmic_roi_bytes(h.sorted_rois[:k].cpu().numpy(), h.roi_level[:k].cpu().numpy(), syn.fpn_level_shapes(shape)[:4], shape, 256)
    rows.append((a['B_roi']/1e6, a['unique_cells'], np.bincount(h.roi_level[:k].cpu().numpy(), minlength=4).tolist()))
for r in rows: print(r)
b=np.array([r[0] for r in rows]); print('B_roi MB mean %.1f max %.1f min %.1f max/mean %.3f'%(b.mean(),b.max(),b.min(),b.max()/b.mean()))
