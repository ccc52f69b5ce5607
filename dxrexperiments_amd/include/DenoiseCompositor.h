// DenoiseCompositor.h -- mirror of the reference class (include/DenoiseCompositor.h:5-59,
// src/DenoiseCompositor.cpp) over the C ABI.
#pragma once

#include "DXRFramework.h"

class DenoiseCompositor
{
public:
    using SharedPtr = std::shared_ptr<DenoiseCompositor>;

    static SharedPtr create(DXRFramework::RtContext::SharedPtr context) { return SharedPtr(new DenoiseCompositor(context)); }
    ~DenoiseCompositor() { rt_denoiser_destroy(mDenoiser); }

    void userInterface() {}

    struct InputComponents      // SRV handles in the reference: device pointers of RGBA32F images here
    {
        const void *directLightingSrv;
        const void *indirectSpecularSrv;
    };

    typedef rt_denoiser_params DenoiserParams;     // DenoiseCompositor.h:41-49
    DenoiserParams &params() { DenoiserParams *p = nullptr; DXRFramework::ThrowIfFailed(rt_denoiser_get_params(mDenoiser, &p)); return *p; }

    void dispatch(InputComponents inputs, unsigned frameIndex, unsigned width, unsigned height)
    {
        (void)frameIndex;
        DXRFramework::ThrowIfFailed(rt_denoiser_dispatch(mDenoiser, inputs.directLightingSrv, inputs.indirectSpecularSrv, width, height));
    }
    void loadResources(unsigned frameCount, bool loadMockResources) { (void)frameCount; (void)loadMockResources; }
    void createOutputResource(unsigned format, unsigned width, unsigned height) { DXRFramework::ThrowIfFailed(rt_denoiser_create_output(mDenoiser, format, width, height)); }
    void *getOutputResource() { void *p = nullptr; DXRFramework::ThrowIfFailed(rt_denoiser_get_output_device_ptr(mDenoiser, &p)); return p; }
    void readOutput(void *host, size_t bytes) { DXRFramework::ThrowIfFailed(rt_denoiser_read_output(mDenoiser, host, bytes)); }

    bool mActive = true;

private:
    explicit DenoiseCompositor(DXRFramework::RtContext::SharedPtr context) : mRtContext(context)
    {
        DXRFramework::ThrowIfFailed(rt_denoiser_create(context->getHandle(), &mDenoiser));
    }
    DXRFramework::RtContext::SharedPtr mRtContext;
    rt_denoiser *mDenoiser = nullptr;
};
