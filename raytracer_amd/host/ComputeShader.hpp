// Forwarding header: the class of this name (reference include/ComputeShader.hpp) lives in render.hpp.
#include "render.hpp"
