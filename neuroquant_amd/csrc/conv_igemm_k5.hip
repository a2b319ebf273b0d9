#define NQ_KS 5
#include "conv_igemm_impl.h"
