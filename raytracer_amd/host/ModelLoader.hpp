// Forwarding header: the class of this name (reference include/ModelLoader.hpp) lives in scene.hpp.
#include "scene.hpp"
