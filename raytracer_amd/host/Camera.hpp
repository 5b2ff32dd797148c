// Forwarding header: the class of this name (reference include/Camera.hpp) lives in render.hpp.
#include "render.hpp"
