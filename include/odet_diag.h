/* odet_diag.h -- entry points that exist ONLY in the diagnostic build of the library (-DODET_DIAG:
 * tools/libodet_hip_diag.so, built by tf_eager_object_detection_amd/_build.py build_diag() and loaded explicitly by tools/ and by
 * the forced-tile tests through tools/_diag.py).  The shipped libodet_hip.so exports none of them and holds no process-global
 * override: what a launch does there is a function of its arguments only.  No reference counterpart. */
#ifndef ODET_DIAG_H_
#define ODET_DIAG_H_
#include "odet.h"

#ifdef __cplusplus
extern "C" {
#endif

/* forces the workgroup tile of this process's next float16 3x3 (form 0; the fused bottleneck tail included) / pointwise (form 1)
 * launches -- {nw waves, wn waves along the channels, mt 16-pixel tiles per wave, ns LDS stages}: (nw / wn) * 16 * mt pixels x
 * 64 * wn channels; ns == 2 the half-step-pipelined loop, ns > 2 the ring forms for launches with few pixels.  nw = 0 clears. */
int odet_debug_conv_tile(int form, int nw, int wn, int mt, int ns);
/* the same for the split-precision float32 launches (csrc/conv_x3.hip): (mt, wn) of its tile list and the K split (workgroups
 * per tile, 1 = none); mt = 0 clears */
int odet_debug_x3_tile(int mt, int wn, int ksplit);

#ifdef __cplusplus
}
#endif
#endif /* ODET_DIAG_H_ */
