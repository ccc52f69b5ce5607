// RealtimeRaytracingPipeline.h -- mirror of the reference class (include/RealtimeRaytracingPipeline.h:15-74,
// src/RealtimeRaytracingPipeline.cpp) over the C ABI: 1 spp, no accumulation, two outputs
// (0 = direct lighting, 1 = indirect specular) that feed the DenoiseCompositor.
#pragma once

#include <vector>

#include "RaytracingPipeline.h"

class RealtimeRaytracingPipeline : public RaytracingPipeline
{
public:
    using SharedPtr = std::shared_ptr<RealtimeRaytracingPipeline>;

    static SharedPtr create(DXRFramework::RtContext::SharedPtr context, uint32_t rngSeed = 1234) { return SharedPtr(new RealtimeRaytracingPipeline(context, rngSeed)); }
    virtual ~RealtimeRaytracingPipeline()
    {
        rt_progressive_host_destroy(mHost);
        rt_pipeline_destroy(mPipeline);
    }

    virtual void userInterface() override {}

    virtual void update(float elapsedTime, UINT elapsedFrames, UINT prevFrameIndex, UINT frameIndex, UINT width, UINT height) override
    {
        (void)prevFrameIndex; (void)frameIndex;
        float cam[11];
        mCamera->Pack(cam);
        DXRFramework::ThrowIfFailed(rt_progressive_host_set_flags(mHost, 1, mAnimationPaused));
        DXRFramework::ThrowIfFailed(rt_realtime_host_update(mHost, cam, elapsedTime, elapsedFrames, width, height, &mConstants));
        DXRFramework::ThrowIfFailed(rt_pipeline_update(mPipeline, &mConstants));
    }

    using RaytracingPipeline::render;       // the (commandList, frameIndex, width, height) form of the reference
    virtual void render(UINT frameIndex, UINT width, UINT height) override
    {
        (void)frameIndex;
        auto program = mRtBindings->getProgram();
        for (UINT rayType = 0; rayType < program->getHitProgramCount(); ++rayType) {
            for (UINT instance = 0; instance < mRtScene->getNumInstances(); ++instance) {
                auto &hitVars = mRtBindings->getHitVars(rayType, instance);
                hitVars->appendHeapRanges(0);
                hitVars->appendHeapRanges(0);
                const Material &m = mMaterials[instance < mMaterials.size() ? instance : mMaterials.size() - 1];
                hitVars->append32BitConstants(&m.params, sizeof(MaterialParams) / 4);
            }
        }
        for (UINT rayType = 0; rayType < program->getMissProgramCount(); ++rayType) {
            auto &missVars = mRtBindings->getMissVars(rayType);
            missVars->appendHeapRanges(0);
            missVars->appendHeapRanges(0);
        }
        mRtBindings->apply(mRtContext, mRtState);
        mRtContext->raytrace(mRtBindings, mRtState, width, height, 3);
    }

    virtual void loadResources(UINT frameCount) override { (void)frameCount; }
    void loadEnvironmentDDS(const std::string &path) { DXRFramework::ThrowIfFailed(rt_pipeline_load_environment_dds(mPipeline, path.c_str())); }
    void setEnvironmentCube(const float *facesRGBA32F, uint32_t size) { DXRFramework::ThrowIfFailed(rt_pipeline_set_environment_cube(mPipeline, facesRGBA32F, size)); }
    void setEnvironmentConstant(float r, float g, float b) { const float c[3] = {r, g, b}; DXRFramework::ThrowIfFailed(rt_pipeline_set_environment_constant(mPipeline, c)); }

    virtual void createOutputResource(UINT format, UINT width, UINT height) override { DXRFramework::ThrowIfFailed(rt_pipeline_create_output(mPipeline, format, width, height)); }
    virtual void buildAccelerationStructures() override { mRtScene->build(mRtContext, mRtProgram->getHitProgramCount()); }

    virtual void addMaterial(Material material) override { mMaterials.push_back(material); }
    virtual void setCamera(std::shared_ptr<Math::Camera> camera) override { mCamera = camera; }
    virtual void setScene(DXRFramework::RtScene::SharedPtr scene) override
    {
        mRtScene = scene;
        mRtBindings = DXRFramework::RtBindings::create(mRtContext, mRtProgram, scene);
        mRtBindings->bindPipeline(mPipeline);
        DXRFramework::ThrowIfFailed(rt_pipeline_set_scene(mPipeline, scene->getHandle(mRtContext)));
    }

    virtual int getNumOutputs() override { return kNumOutputResources; }
    virtual void *getOutputResource(UINT id) override { void *p = nullptr; DXRFramework::ThrowIfFailed(rt_pipeline_get_output_device_ptr(mPipeline, id, &p)); return p; }
    void readOutput(UINT id, void *host, size_t bytes) { DXRFramework::ThrowIfFailed(rt_pipeline_read_output_n(mPipeline, id, host, bytes)); }

    virtual bool *isActive() override { return &mActive; }
    virtual const char *getName() override { return rt_pipeline_get_name(mPipeline); }
    void setAnimationPaused(bool on) { mAnimationPaused = on; }

private:
    RealtimeRaytracingPipeline(DXRFramework::RtContext::SharedPtr context, uint32_t rngSeed) : mRtContext(context)
    {
        using namespace DXRFramework;
        RtProgram::Desc programDesc;          // src/RealtimeRaytracingPipeline.cpp:33-39
        programDesc.addShaderLibrary({L"RayGen", L"PrimaryClosestHit", L"PrimaryMiss", L"ShadowClosestHit", L"ShadowAnyHit", L"ShadowMiss"});
        programDesc.setRayGen("RayGen");
        programDesc.addHitGroup(0, "PrimaryClosestHit", "").addMiss(0, "PrimaryMiss");
        programDesc.addHitGroup(1, "ShadowClosestHit", "ShadowAnyHit").addMiss(1, "ShadowMiss");
        mRtProgram = RtProgram::create(context, programDesc);
        mRtState = RtState::create(context);
        mRtState->setProgram(mRtProgram);
        mRtState->setMaxTraceRecursionDepth(4);
        mRtState->setMaxAttributeSize(8);
        mRtState->setMaxPayloadSize(60);      // RealtimePayload (:69-71)
        ThrowIfFailed(rt_pipeline_create(context->getHandle(), RT_PIPELINE_REALTIME, &mPipeline));
        ThrowIfFailed(rt_progressive_host_create(rngSeed, &mHost));
        std::memset(&mConstants, 0, sizeof mConstants);
    }

    const int kNumOutputResources = 2;
    DXRFramework::RtContext::SharedPtr mRtContext;
    DXRFramework::RtProgram::SharedPtr mRtProgram;
    DXRFramework::RtBindings::SharedPtr mRtBindings;
    DXRFramework::RtState::SharedPtr mRtState;
    DXRFramework::RtScene::SharedPtr mRtScene;
    std::vector<Material> mMaterials;
    std::shared_ptr<Math::Camera> mCamera;
    rt_pipeline *mPipeline = nullptr;
    rt_progressive_host *mHost = nullptr;
    PerFrameConstants mConstants;
    bool mActive = true, mAnimationPaused = true;
};
