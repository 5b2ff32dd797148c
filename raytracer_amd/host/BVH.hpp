// Forwarding header: the class of this name (reference include/BVH.hpp) lives in scene.hpp.
#include "scene.hpp"
