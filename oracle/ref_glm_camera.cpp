// ref_glm_camera.cpp — TEST INFRASTRUCTURE.  Pins raytracer_amd/host/rtmath.hpp against the
// reference's vendored glm 0.9.9.7 (compiled from where it lies, -I$(REFDIR)/include).
// Performs the arithmetic of Camera::get{View,Perspective}Matrix (reference src/Camera.cpp:9-17)
// and Tracer::updateRays (reference src/Tracer.cpp:110-130) with real glm calls and prints the
// four corner rays as hex floats.  Camera.cpp/Tracer.cpp themselves need glad/GLFW headers the
// image lacks, so they are not compiled; only glm is.
//   usage: ref_glm_camera ex ey ez lx ly lz ux uy uz fov w h near far
#include <cstdio>
#include <cstdlib>
#include <glm/glm/mat4x4.hpp>
#include <glm/glm/vec4.hpp>
#include <glm/glm/gtc/matrix_transform.hpp>

int main(int argc, char **argv) {
    if (argc != 15) return 2;
    float a[14];
    for (int i = 0; i < 14; i++) a[i] = strtof(argv[i + 1], nullptr);
    glm::vec3 eye(a[0], a[1], a[2]), lookAt(a[3], a[4], a[5]), up(a[6], a[7], a[8]);
    float fov = a[9], x = a[10], y = a[11], zn = a[12], zf = a[13];
    glm::mat4 m = glm::perspective(glm::radians(fov), x / y, zn, zf) * glm::lookAt(eye, lookAt, up);
    m = glm::transpose(glm::inverse(m));
    const float c[4][2] = {{-1, -1}, {-1, +1}, {+1, -1}, {+1, +1}};
    for (int k = 0; k < 4; k++) {
        glm::vec4 r = glm::vec4(c[k][0], c[k][1], 0, 1) * m;
        r /= r.w;
        r -= glm::vec4(eye, 0.0f);
        printf("%a %a %a\n", r.x, r.y, r.z);
    }
    return 0;
}
