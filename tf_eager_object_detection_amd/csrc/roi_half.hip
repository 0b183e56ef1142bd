// The float16 instantiations of k_roi_pool (see the note in roi.hip): the same source, compiled WITH hipcc's SLP
// vectoriser (no -fno-slp-vectorize for this file in _build.PER_SOURCE_FLAGS).
#define ODET_ROI_HALF_TU 1
#include "roi.hip"
