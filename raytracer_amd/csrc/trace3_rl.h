// trace3_rl.h — launch interface of the Q-table guided kernel (trace3_rl_kernel.hip).
#ifndef RT_TRACE3_RL_H
#define RT_TRACE3_RL_H

#include "rt_device.h"

struct RtQTable { // qtablehelper.glsl:18-32 (bindings 7..10) and its uniforms numPoints / maxTheta / maxPhi
    const float4 *normals, *centres; // vec3[] with the std430 stride of 16 bytes
    const float4 *cells; // derived (rtPrepareCells): [2i] = cell i's normal normalised, w = 1 if it is non-zero; [2i+1] = centre
    float *qt;
    int *visits;
    int numNormals, numCentres, numQ, numVisits;
    int numPoints, maxTheta, maxPhi;
};

hipError_t rtLaunchTrace3RL(const RtScene &sc, const RtQTable &q, const RtImages &img, const rt_trace_params &p,
                            const rt_rect &tile, RtCountersDev *counters, hipStream_t stream);

// fills out[0 .. 2 * q.numNormals) from q.normals / q.centres (a centre beyond q.numCentres reads as zero)
hipError_t rtPrepareCells(const RtQTable &q, float4 *out, hipStream_t stream);

#endif
