// Forwarding header: the class of this name (reference include/QTable.hpp) lives in scene.hpp.
#include "scene.hpp"
