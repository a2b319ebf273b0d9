#define NQ_KS 1
#include "conv_wgrad_impl.h"
