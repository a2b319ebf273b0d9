#define NQ_KS 3
#include "conv_igemm_impl.h"
