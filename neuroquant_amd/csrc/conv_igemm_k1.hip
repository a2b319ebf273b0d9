#define NQ_KS 1
#include "conv_igemm_impl.h"
