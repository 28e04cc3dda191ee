// Entry points declared in include/sdmi.h whose kernels are not written yet.  They fail loudly
// (SDMI_EUNSUPPORTED) -- no silent fallback.  Each one moves to its own .hip file when implemented.
#include "../../include/sdmi.h"
void sdmi_set_error(const char* fmt, ...);
#define PENDING(name, T)                                         \
  extern "C" int name(const T*, void*) {                         \
    sdmi_set_error(#name ": not implemented in this build");    \
    return SDMI_EUNSUPPORTED;                                    \
  }
PENDING(sdmi_wgrad, SdmiWgradArgs)
PENDING(sdmi_groupnorm_bwd, SdmiGroupNormBwdArgs)
PENDING(sdmi_layernorm_bwd, SdmiLayerNormBwdArgs)
PENDING(sdmi_attention_bwd, SdmiAttnBwdArgs)
PENDING(sdmi_sqsum_partial, SdmiSqSumArgs)
PENDING(sdmi_adam_clip, SdmiAdamArgs)
