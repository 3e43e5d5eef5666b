// Library identification (include/tmae_hip.h): the hand-bumped ABI version and the signature fingerprint of the header this
// object was compiled against (build.py computes it with tmae_amd/_abi.py and passes -DTMAE_ABI_HASH; _lib.py recomputes it from
// its ctypes table at import and refuses a library whose value differs).
#include "common.h"

#ifndef TMAE_ABI_HASH
#error "build with t-mae_amd/build.py: it passes -DTMAE_ABI_HASH=<fingerprint of include/tmae_hip.h>"
#endif

int tmae_abi_version(void) { return TMAE_ABI_VERSION; }
int tmae_abi_hash(void) { return TMAE_ABI_HASH; }
