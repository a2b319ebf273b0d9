#define NQ_KS 3
#include "conv_wgrad_impl.h"
