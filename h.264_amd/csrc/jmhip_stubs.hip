// jmhip_stubs.hip -- entry points declared in include/jmhip.h whose device stage is not built yet:
// they fail loudly (never a CPU fallback).
#include "jmhip_internal.h"

