// Device helpers shared by the matcher kernels.
#pragma once
#include "common.h"

namespace orbgpu {

// ORBmatcher::ComputeThreeMaxima, ORBmatcher.cc:1601-1642
__device__ inline void three_maxima(const int *histo, int L, int &ind1, int &ind2, int &ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2;
            max2 = max1;
            max1 = s;
            ind3 = ind2;
            ind2 = ind1;
            ind1 = i;
        } else if (s > max2) {
            max3 = max2;
            max2 = s;
            ind3 = ind2;
            ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if ((float)max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

// rotation bin, ORBmatcher.cc:238-243
__device__ __forceinline__ int rot_bin(float angle_a, float angle_b)
{
    const float factor = 1.0f / ORBGPU_HISTO_LENGTH;
    float rot = angle_a - angle_b;
    if (rot < 0.0f)
        rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == ORBGPU_HISTO_LENGTH)
        bin = 0;
    return bin;
}

} // namespace orbgpu
