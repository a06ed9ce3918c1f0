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
// pointer (or nullptr: skip the display transform); errors abort like checkCUDAError
// ("HIP error (file:line): msg: string" on stderr, exit(EXIT_FAILURE)).
// One difference, allowed by SURVEY 8(b) "Outputs" (the per-iteration D2H may be lazy as long as state.image is current when saveImage() runs):
// with pbo == nullptr pathtrace() only ENQUEUES the iteration — no host wait, no 7.7 MB copy of the radiance sum after every call
// (pathtrace.cu:809-810) — and scene->state.image / an exact guiData->TracedDepth are made current by the call with iteration >=
// scene->state.iterations (the one after which main.cpp:221-222 saves and exits), by pathtraceFree(), and by pathtraceResolve() for a headless
// caller that reads them earlier.  With a pbo (the viewer, where the S key may save at any time) every call refreshes them, as the reference does.
// An unedited main.cpp is therefore correct by default.  PTK_EAGER_IMAGE=1 / PTK_LAZY_IMAGE=1 in the environment force either behaviour.
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
void pathtraceResolve();            // not in the reference: makes scene->state.image / TracedDepth current (see above)
