// placeholder until the LunarLander kernels land
#include "common.h"
namespace mgym { Env* make_lunarlander() { return nullptr; } }
