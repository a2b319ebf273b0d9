#define NQ_KS 3
#include "conv_igemm3_impl.h"
