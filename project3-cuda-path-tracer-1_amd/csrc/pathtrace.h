// pathtrace.h — the reference's render API (src/pathtrace.h:6-9) with the same four signatures.
//
//   void InitDataContainer(GuiDataContainer* guiData);
//   void pathtraceInit(Scene* scene);
//   void pathtraceFree();
//   void pathtrace(uchar4* pbo, int frame, int iteration);
void pathtraceResolve();            // not in the reference: makes scene->state.image / TracedDepth current (see above)
//
// Behaviour kept from src/pathtrace.cu: file-static state, one scene at a time, not re-entrant;
// pathtraceFree() is null-safe and may be called before any pathtraceInit() and twice in a row
// (main.cpp:198-201 does exactly that); `iteration` is 1-based; `frame` is unused; `pbo` is a DEVICE
// pointer (or nullptr: skip the display transform); errors abort like checkCUDAError
// ("HIP error (file:line): msg: string" on stderr, exit(EXIT_FAILURE)).
// One difference, allowed by SURVEY 8(b) "Outputs": pathtrace() only ENQUEUES the iteration — it does not wait for the GPU and does
// not copy the 7.7 MB radiance sum to the host after every call (pathtrace.cu:809-810).  scene->state.image and an exact
// guiData->TracedDepth are produced by pathtraceResolve(), which the caller runs before it reads them (saveImage(), main.cpp:146-170);
// between resolves TracedDepth follows from an asynchronous read-back, a few calls late.  PTK_EAGER_IMAGE=1 in the environment restores
// the copy after every call for a caller that cannot be edited.
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
