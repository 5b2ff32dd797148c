// Forwarding header: the class of this name (reference include/filter.hpp) lives in render.hpp.
#include "render.hpp"
