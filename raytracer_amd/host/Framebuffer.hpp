// Forwarding header: the class of this name (reference include/Framebuffer.hpp) lives in render.hpp.
#include "render.hpp"
