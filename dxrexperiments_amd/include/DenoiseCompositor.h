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

    struct InputComponents      // DenoiseCompositor.h:34-38: SRV handles; a handle's ptr is the device address of an RGBA32F image
    {
        D3D12_GPU_DESCRIPTOR_HANDLE directLightingSrv;
        D3D12_GPU_DESCRIPTOR_HANDLE indirectSpecularSrv;
    };
    static D3D12_GPU_DESCRIPTOR_HANDLE handleOf(const void *deviceImage) { return D3D12_GPU_DESCRIPTOR_HANDLE{(unsigned long long)(size_t)deviceImage}; }

    typedef rt_denoiser_params DenoiserParams;     // DenoiseCompositor.h:41-49
    DenoiserParams &params() { DenoiserParams *p = nullptr; DXRFramework::ThrowIfFailed(rt_denoiser_get_params(mDenoiser, &p)); return *p; }

    void dispatch(InputComponents inputs, unsigned frameIndex, unsigned width, unsigned height)
    {
        (void)frameIndex;
        DXRFramework::ThrowIfFailed(rt_denoiser_dispatch(mDenoiser, (const void *)(size_t)inputs.directLightingSrv.ptr,
                                                         (const void *)(size_t)inputs.indirectSpecularSrv.ptr, width, height));
    }
    // the reference's signature (DenoiseCompositor.h:20, call site src/DXRExperimentsApp.cpp:206); the command list is ignored
    void dispatch(ID3D12GraphicsCommandList *, InputComponents inputs, unsigned frameIndex, unsigned width, unsigned height)
    {
        dispatch(inputs, frameIndex, width, height);
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
