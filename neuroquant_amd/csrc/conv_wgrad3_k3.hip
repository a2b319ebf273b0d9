#define NQ_KS 3
#include "conv_wgrad3_impl.h"
