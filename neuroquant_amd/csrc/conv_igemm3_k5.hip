#define NQ_KS 5
#include "conv_igemm3_impl.h"
