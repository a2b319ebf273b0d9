#define NQ_KS 5
#include "conv_wgrad_impl.h"
