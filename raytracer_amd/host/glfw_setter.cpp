// glfw_setter.cpp — see glfw_setter.hpp.
#include "glfw_setter.hpp"

#include <algorithm>
#include <cmath>

using rtm::vec3;

FrameBuffer *GlfwSetter::framebuffer = nullptr;
Camera *GlfwSetter::camera = nullptr;
int GlfwSetter::frameNumber = 0;
float GlfwSetter::distMoved = 0;
float GlfwSetter::camHeight = 3.0f;
bool *GlfwSetter::importanceSampled = nullptr;
bool *GlfwSetter::useBlueNoise = nullptr;
bool *GlfwSetter::accumulateSamples = nullptr;
bool *GlfwSetter::freezeTime = nullptr;
int *GlfwSetter::maxAccumulateSamples = nullptr;
int *GlfwSetter::iterations = nullptr;
int *GlfwSetter::maxBounces = nullptr;
bool GlfwSetter::filter = false;
float GlfwSetter::exposure = 1.0f;
bool GlfwSetter::extendedKeys = false;
int GlfwSetter::lastX = 0;
int GlfwSetter::lastY = 0;
float GlfwSetter::yaw = 0;
float GlfwSetter::pitch = 0;
bool GlfwSetter::firstMouse = true;

void GlfwSetter::reset() {
    framebuffer = nullptr;
    camera = nullptr;
    frameNumber = 0;
    distMoved = 0;
    camHeight = 3.0f;
    importanceSampled = useBlueNoise = accumulateSamples = freezeTime = nullptr;
    maxAccumulateSamples = iterations = maxBounces = nullptr;
    filter = false;
    exposure = 1.0f;
    lastX = lastY = 0;
    yaw = pitch = 0;
    firstMouse = true;
}

namespace {

vec3 normalized(vec3 v) {
    const float l = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return vec3(v.x / l, v.y / l, v.z / l);
}
vec3 crossOf(vec3 a, vec3 b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

// W / A / S / D (glfw_setter.cpp:83-154): a 0.1 step in the horizontal plane, clamped to the room (|x|, |z| <= 5)
void moveHorizontal(Camera *cam, vec3 eyediff, float sign, float &distMoved) {
    eyediff.y = 0;
    if (cam->eye.x + sign * eyediff.x > 5.0f || cam->eye.x + sign * eyediff.x < -5.0f) eyediff.x = 0;
    if (cam->eye.z + sign * eyediff.z > 5.0f || cam->eye.z + sign * eyediff.z < -5.0f) eyediff.z = 0;
    cam->eye = cam->eye + eyediff * sign;
    cam->lookAt = cam->lookAt + eyediff * sign;
    distMoved += std::sqrt(eyediff.x * eyediff.x + eyediff.y * eyediff.y + eyediff.z * eyediff.z);
}

} // namespace

void GlfwSetter::key_callback(int key, int, int action, int) {
    if (!camera) return;
    if (key == GLFW_KEY_W || key == GLFW_KEY_S) {
        moveHorizontal(camera, normalized(camera->lookAt - camera->eye) * 0.1f, key == GLFW_KEY_W ? 1.0f : -1.0f, distMoved);
        frameNumber = 0;
    }
    if (key == GLFW_KEY_A || key == GLFW_KEY_D) {
        moveHorizontal(camera, normalized(crossOf(camera->lookAt - camera->eye, camera->up)) * 0.1f,
                       key == GLFW_KEY_D ? 1.0f : -1.0f, distMoved);
        frameNumber = 0;
    }
    if (key == GLFW_KEY_Z || key == GLFW_KEY_X) { // :155-176: up / down, not below y = 0
        vec3 eyediff = normalized(camera->up) * 0.1f;
        eyediff.x = eyediff.z = 0;
        if (key == GLFW_KEY_X && camera->eye.y - eyediff.y < 0.0f) eyediff.y = 0;
        const float sign = key == GLFW_KEY_Z ? 1.0f : -1.0f;
        camera->eye = camera->eye + eyediff * sign;
        camera->lookAt = camera->lookAt + eyediff * sign;
        frameNumber = 0;
    }
    if (extendedKeys && action == GLFW_PRESS) { // :177-206, commented out in the reference
        if (key == GLFW_KEY_C && importanceSampled) { *importanceSampled = !*importanceSampled; frameNumber = 0; }
        if (key == GLFW_KEY_B && useBlueNoise) { *useBlueNoise = !*useBlueNoise; frameNumber = 0; }
        if (key == GLFW_KEY_F && accumulateSamples) { *accumulateSamples = !*accumulateSamples; frameNumber = 0; }
        if (key == GLFW_KEY_T && freezeTime) { *freezeTime = !*freezeTime; frameNumber = 0; }
        if (key == GLFW_KEY_UP && maxAccumulateSamples) { (*maxAccumulateSamples)++; frameNumber = 0; }
        if (key == GLFW_KEY_DOWN && maxAccumulateSamples) {
            *maxAccumulateSamples = std::max(1, *maxAccumulateSamples - 1);
            frameNumber = 0;
        }
    }
    if (key == GLFW_KEY_MINUS && action == GLFW_PRESS && iterations) *iterations = std::max(*iterations - 1, 1);
    if (key == GLFW_KEY_EQUAL && action == GLFW_PRESS && iterations) (*iterations)++;
    if (key == GLFW_KEY_SPACE && action == GLFW_PRESS) filter = !filter;
    if (key == GLFW_KEY_COMMA && action == GLFW_PRESS && maxBounces) { (*maxBounces)++; frameNumber = 0; }
    if (key == GLFW_KEY_PERIOD && action == GLFW_PRESS && maxBounces) {
        *maxBounces = std::max(2, *maxBounces - 1);
        frameNumber = 0;
    }
    if (key == GLFW_KEY_1) exposure += 0.1f;
    if (key == GLFW_KEY_2) exposure = (float)std::max(0.0, (double)exposure - 0.1);
}

void GlfwSetter::scroll_callback(double, double yoffset) {
    if (!camera) return;
    const float tempfov = camera->fov;
    camera->fov = (float)std::max(std::min((double)camera->fov - yoffset, 100.0), 45.0);
    if (camera->fov != tempfov) frameNumber = 0;
}

namespace {

const double kDegPerRad = 57.295779513082320876798154814105; // glm::degrees / glm::radians in double
double toDegrees(double r) { return r * kDegPerRad; }
double toRadians(double d) { return d * 0.01745329251994329576923690768489; }

} // namespace

void GlfwSetter::mouse_callback(double xpos, double ypos) {
    if (!camera) return;
    if (firstMouse) {
        // the angles of the current viewing direction: pitch from y; yaw from asin(z) and acos(x) of the horizontal part,
        // pieced together per quadrant into (-90, 270] degrees (glfw_setter.cpp:246-280)
        lastX = (int)xpos;
        lastY = (int)ypos;
        const vec3 d = normalized(camera->lookAt - camera->eye);
        const double p = toDegrees(std::asin(d.y));
        pitch = (float)p;
        const double horizontal = std::cos(toRadians(p));
        const double fromZ = toDegrees(std::asin(d.z / horizontal)), fromX = toDegrees(std::acos(d.x / horizontal));
        const bool front = fromX < 90.0; // x > 0
        yaw = (float)(fromZ > 0 ? (front ? fromZ : fromX) : (front ? fromZ : 180.0 - fromZ));
        firstMouse = false;
        return;
    }
    const float sensitivity = 0.05f;
    const float dx = (float)(xpos - lastX) * sensitivity, dy = (float)(lastY - ypos) * sensitivity;
    lastX = (int)xpos;
    lastY = (int)ypos;
    const float turned = yaw + dx;
    yaw = turned - 360.0f * std::floor(turned / 360.0f); // glm::mod
    pitch = std::min(89.0f, std::max(-89.0f, pitch + dy));
    const float yr = yaw * 0.01745329251994329576923690768489f, pr = pitch * 0.01745329251994329576923690768489f;
    const vec3 direction(std::cos(yr) * std::cos(pr), std::sin(pr), std::sin(yr) * std::cos(pr));
    camera->lookAt = camera->eye + normalized(direction);
    frameNumber = 0;
}
