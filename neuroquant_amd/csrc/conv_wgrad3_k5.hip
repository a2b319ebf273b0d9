#define NQ_KS 5
#include "conv_wgrad3_impl.h"
