// analytic.h — launch interface of the analytic-scene kernels (analytic_kernels.hip).
#ifndef RT_ANALYTIC_H
#define RT_ANALYTIC_H

#include "rt_device.h"

#define AN_MAX_BATCH 16

struct AnalyticSample { // what differs between the dispatches of one batch
    float time, blendFactor;
    int frameIndex, pad;
};

struct AnalyticParams { // uniforms of trace.cs:7-21, trace1.cs:14-34, trace2.cs:10-11
    float eye[3], ray00[3], ray01[3], ray10[3], ray11[3];
    int importanceSampled, multipleImportanceSampled, useBlueNoise;
    float phongExponent, specularFactor, roughness;
    int flags, batch;
    AnalyticSample s[AN_MAX_BATCH];
};

struct AnalyticTables { // rt_set_table contents; only the ones a kernel reads need to be set
    const uint8_t *sobol, *scramble, *ranking; // trace1.cs:83-93 (65536 / 131072 / 131072 bytes)
    const float4 *ltc1, *ltc2;                 // trace2.cs:12-13 (64x64 RGBA32F)
};

hipError_t rtLaunchAnalytic(int kernel, const RtImages &img, const AnalyticParams &p, const AnalyticTables &tb,
                            const rt_rect &tile, RtCountersDev *counters, hipStream_t stream);

// RT_KERNEL_TRACE3_LTC: `lights` = numRects x nine floats (c, x, y) in device memory
hipError_t rtLaunchTrace3Ltc(const RtScene &sc, const RtImages &img, const AnalyticParams &p, const AnalyticTables &tb,
                             const float *lights, int numRects, float intensity, int maxFollows, const rt_rect &tile,
                             RtCountersDev *counters, hipStream_t stream);

#endif
