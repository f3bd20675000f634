// libnbx: the instances of jk_mx.hip for N = 272 .. 400 (a translation unit of their own: they build beside the others)
#define NBX_MX_SIZES(X) X(68) X(72) X(76) X(80) X(84) X(88) X(92) X(96) X(100)
#define MX_FN(name) name##_hi
#include "jk_mx.hip"
