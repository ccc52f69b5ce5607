// RaytracingPipeline.h -- the abstract pipeline interface of the reference
// (include/RaytracingPipeline.h:8-39) with D3D12 types replaced:
//   ID3D12GraphicsCommandList* / ID3D12CommandQueue*  -> dropped (work goes to the RtContext's stream)
//   DXGI_FORMAT                                       -> RT_FORMAT_* (same numeric values)
//   ID3D12Resource*                                   -> device pointer of the output image
//   D3D12_GPU_DESCRIPTOR_HANDLE (UAV / SRV handles)   -> kept as a type; its `ptr` is the device address of the image
#pragma once

#include <memory>

#include "Camera.h"
#include "DXRFramework.h"

using UINT = unsigned int;
typedef rt_material_params MaterialParams;          // RaytracingHlslCompat.h:87-96
typedef rt_per_frame_constants PerFrameConstants;   // RaytracingHlslCompat.h:79-85
typedef rt_debug_options DebugOptions;              // RaytracingHlslCompat.h:64-77

class RaytracingPipeline
{
public:
    using SharedPtr = std::shared_ptr<RaytracingPipeline>;
    virtual ~RaytracingPipeline() {}

    virtual void userInterface() = 0;
    virtual void update(float elapsedTime, UINT elapsedFrames, UINT prevFrameIndex, UINT frameIndex, UINT width, UINT height) = 0;
    virtual void render(UINT frameIndex, UINT width, UINT height) = 0;
    // the reference's signature (include/RaytracingPipeline.h:18); the command list is ignored
    void render(ID3D12GraphicsCommandList *, UINT frameIndex, UINT width, UINT height) { render(frameIndex, width, height); }

    virtual void loadResources(UINT frameCount) = 0;
    virtual void createOutputResource(UINT format, UINT width, UINT height) = 0;
    virtual void buildAccelerationStructures() = 0;

    struct Material
    {
        MaterialParams params;
    };

    virtual void addMaterial(Material material) = 0;
    virtual void setCamera(std::shared_ptr<Math::Camera> camera) = 0;
    virtual void setScene(DXRFramework::RtScene::SharedPtr scene) = 0;

    virtual int getNumOutputs() = 0;
    virtual void *getOutputResource(UINT id) = 0;
    // include/RaytracingPipeline.h:35-36: what DenoiseCompositor::dispatch is fed (src/DXRExperimentsApp.cpp:202-206)
    virtual D3D12_GPU_DESCRIPTOR_HANDLE getOutputUavHandle(UINT id) { return D3D12_GPU_DESCRIPTOR_HANDLE{(unsigned long long)(size_t)getOutputResource(id)}; }
    virtual D3D12_GPU_DESCRIPTOR_HANDLE getOutputSrvHandle(UINT id) { return D3D12_GPU_DESCRIPTOR_HANDLE{(unsigned long long)(size_t)getOutputResource(id)}; }

    virtual bool *isActive() = 0;
    virtual const char *getName() = 0;
};
