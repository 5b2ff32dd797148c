// Forwarding header: the class of this name (reference include/Tracer.hpp) lives in render.hpp.
#include "render.hpp"
