"""tf_eager_object_detection_amd -- MI355X-native Faster-R-CNN / FPN detection hot path.

Drop-in counterparts of the reference's ``object_detection.model.{region_proposal,roi_pooling,
prediction}`` and ``object_detection.utils.{anchor_generator,bbox_tf,bbox_transform}`` modules,
implemented as hand-written HIP kernels (gfx950) behind a C ABI (include/odet.h).  Tensors are
PyTorch-ROCm GPU tensors (NHWC feature maps, [x1,y1,x2,y2] float32 boxes).
"""
__version__ = '0.1.0'
