// glfw_setter.hpp — the STATE MACHINE of the reference's input handler without the window
// (reference include/glfw_setter.hpp:37-49, src/glfw_setter.cpp:77-237,304-310): which inputs move the camera, which
// reset the progressive accumulation (frameNumber = 0), which toggle the filter / change bounces, iterations, exposure.
// A headless run feeds it scripted events (rt_headless --keys "FRAME:KEY,..."), so that "camera change -> accumulation
// restarts" can be exercised and tested without GLFW.  Mouse-look (src/glfw_setter.cpp:239-302) is here too: a cursor
// position turns the camera and restarts the accumulation.  Window creation, cursor capture and the F10 / ESC bindings
// stay out of scope (component 18).
#ifndef RT_AMD_GLFW_SETTER_HPP
#define RT_AMD_GLFW_SETTER_HPP

#include "render.hpp"

// key codes as GLFW numbers them (printable keys = their ASCII code)
enum {
    GLFW_KEY_SPACE = 32, GLFW_KEY_COMMA = 44, GLFW_KEY_MINUS = 45, GLFW_KEY_PERIOD = 46, GLFW_KEY_1 = 49, GLFW_KEY_2 = 50,
    GLFW_KEY_EQUAL = 61, GLFW_KEY_A = 65, GLFW_KEY_B = 66, GLFW_KEY_C = 67, GLFW_KEY_D = 68, GLFW_KEY_F = 70,
    GLFW_KEY_S = 83, GLFW_KEY_T = 84, GLFW_KEY_W = 87, GLFW_KEY_X = 88, GLFW_KEY_Z = 90, GLFW_KEY_UP = 265,
    GLFW_KEY_DOWN = 264
};
enum { GLFW_RELEASE = 0, GLFW_PRESS = 1, GLFW_REPEAT = 2 };

class GlfwSetter {
public:
    static FrameBuffer *framebuffer;
    static Camera *camera;
    static int frameNumber;
    static float distMoved, camHeight;
    static bool *importanceSampled, *useBlueNoise, *accumulateSamples, *freezeTime;
    static int *maxAccumulateSamples, *iterations, *maxBounces;
    static bool filter;
    static float exposure;

    // glfw_setter.cpp:77-237 without the GLFWwindow argument.  The four bindings that are commented out there
    // (C, B, F, T, UP, DOWN: :177-206) are live here only when `extendedKeys` is set.
    static void key_callback(int key, int scancode, int action, int mods);
    static void scroll_callback(double xoffset, double yoffset); // :304-310
    // :239-302.  The first position only sets the reference point and reads yaw / pitch off the camera; later ones
    // turn the camera by 0.05 degrees per pixel (pitch clamped to +-89) and restart the accumulation.
    static void mouse_callback(double xpos, double ypos);
    static int lastX, lastY; // (whole pixels, as in the reference: reference include/glfw_setter.hpp:25)
    static float yaw, pitch;
    static bool firstMouse;
    static bool extendedKeys;
    static void reset(); // back to the initial values of glfw_setter.cpp:3-23
};

#endif
