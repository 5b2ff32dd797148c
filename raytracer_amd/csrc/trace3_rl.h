// trace3_rl.h — launch interface of the Q-table guided kernel (trace3_rl_kernel.hip).
#ifndef RT_TRACE3_RL_H
#define RT_TRACE3_RL_H

#include "rt_device.h"

struct RtQTable { // qtablehelper.glsl:18-32 (bindings 7..10) and its uniforms numPoints / maxTheta / maxPhi
    const float4 *normals, *centres; // vec3[] with the std430 stride of 16 bytes
    float *qt;
    int *visits;
    int numNormals, numCentres, numQ, numVisits;
    int numPoints, maxTheta, maxPhi;
};

hipError_t rtLaunchTrace3RL(const RtScene &sc, const RtQTable &q, const RtImages &img, const rt_trace_params &p,
                            const rt_rect &tile, RtCountersDev *counters, hipStream_t stream);

#endif
