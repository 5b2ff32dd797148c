// Forwarding header: the class of this name (reference include/Material.hpp) lives in scene.hpp.
#include "scene.hpp"
