// pathtrace.h — the reference's render API (src/pathtrace.h:6-9) with the same four signatures.
//
//   void InitDataContainer(GuiDataContainer* guiData);
//   void pathtraceInit(Scene* scene);
//   void pathtraceFree();
//   void pathtrace(uchar4* pbo, int frame, int iteration);
//
// Behaviour kept from src/pathtrace.cu: file-static state, one scene at a time, not re-entrant;
// pathtraceFree() is null-safe and may be called before any pathtraceInit() and twice in a row
// (main.cpp:198-201 does exactly that); `iteration` is 1-based; `frame` is unused; `pbo` is a DEVICE
// pointer (or nullptr: skip the display transform); after every call scene->state.image holds the
// running radiance SUM; guiData->TracedDepth is updated; errors abort like checkCUDAError
// ("HIP error (file:line): msg: string" on stderr, exit(EXIT_FAILURE)).
// `sampleMode` is the application's global (main.cpp:20); the library carries a weak default (MIS).
#pragma once
#include <hip/hip_runtime.h>
#include "scene.h"

class GuiDataContainer {            // utilities.h:72-77
public:
    GuiDataContainer() : TracedDepth(0) {}
    int TracedDepth;
};

extern SampleMode sampleMode;

void InitDataContainer(GuiDataContainer* guiData);
void pathtraceInit(Scene* scene);
void pathtraceFree();
void pathtrace(uchar4* pbo, int frame, int iteration);
