// ProgressiveRaytracingPipeline.h -- mirror of the reference class
// (include/ProgressiveRaytracingPipeline.h:15-78, src/ProgressiveRaytracingPipeline.cpp) over the C ABI.
// The method bodies follow the reference's order of operations: the constructor describes the program
// and state (:27-89), update() fills the per-frame constants (:177-213), render() re-appends the hit /
// miss record arguments, applies the bindings and dispatches (:215-247).
#pragma once

#include <string>
#include <vector>

#include "RaytracingPipeline.h"

class ProgressiveRaytracingPipeline : public RaytracingPipeline
{
public:
    using SharedPtr = std::shared_ptr<ProgressiveRaytracingPipeline>;

    static SharedPtr create(DXRFramework::RtContext::SharedPtr context, uint32_t rngSeed = 1234) { return SharedPtr(new ProgressiveRaytracingPipeline(context, rngSeed)); }
    virtual ~ProgressiveRaytracingPipeline()
    {
        rt_progressive_host_destroy(mHost);
        rt_pipeline_destroy(mPipeline);
    }

    virtual void userInterface() override {}      // ImGui panels of the reference (:249-312) have no headless equivalent

    virtual void update(float elapsedTime, UINT elapsedFrames, UINT prevFrameIndex, UINT frameIndex, UINT width, UINT height) override
    {
        (void)prevFrameIndex; (void)frameIndex;   // indices into the reference's triple-buffered constant buffer
        float cam[11];
        mCamera->Pack(cam);
        rt_debug_options *opt = nullptr;
        DXRFramework::ThrowIfFailed(rt_progressive_host_options(mHost, &opt));
        *opt = mShaderDebugOptions;
        DXRFramework::ThrowIfFailed(rt_progressive_host_set_flags(mHost, mFrameAccumulationEnabled, mAnimationPaused));
        DXRFramework::ThrowIfFailed(rt_progressive_host_update(mHost, cam, elapsedTime, elapsedFrames, width, height, &mConstants));
        DXRFramework::ThrowIfFailed(rt_pipeline_update(mPipeline, &mConstants));
    }

    using RaytracingPipeline::render;       // the (commandList, frameIndex, width, height) form of the reference
    virtual void render(UINT frameIndex, UINT width, UINT height) override
    {
        (void)frameIndex;
        fillShaderTable();
        mRtContext->raytrace(mRtBindings, mRtState, width, height, 3);
    }

    // The same frame restricted to the interleaved row bands {b : b mod world == rank} of bandRows rows (a tile-partitioned
    // multi-GPU run, examples/progressive_multi.cpp); no counterpart in the single-GPU reference.
    void renderBands(UINT width, UINT height, UINT bandRows, UINT rank, UINT world)
    {
        fillShaderTable();
        DXRFramework::ThrowIfFailed(rt_pipeline_render_bands(mPipeline, width, height, bandRows, rank, world));
    }

    // renderBatch for a tile-partitioned run: n frames of this rank's bands through shared sets of launches
    void renderBandsBatch(float elapsedTime, UINT firstElapsedFrames, UINT n, UINT width, UINT height, UINT bandRows, UINT rank, UINT world)
    {
        std::vector<PerFrameConstants> constants(n);
        float cam[11];
        mCamera->Pack(cam);
        rt_debug_options *opt = nullptr;
        DXRFramework::ThrowIfFailed(rt_progressive_host_options(mHost, &opt));
        *opt = mShaderDebugOptions;
        DXRFramework::ThrowIfFailed(rt_progressive_host_set_flags(mHost, mFrameAccumulationEnabled, mAnimationPaused));
        for (UINT i = 0; i < n; ++i)
            DXRFramework::ThrowIfFailed(rt_progressive_host_update(mHost, cam, elapsedTime, firstElapsedFrames + i, width, height, &constants[i]));
        fillShaderTable();
        DXRFramework::ThrowIfFailed(rt_pipeline_render_bands_batch(mPipeline, width, height, bandRows, rank, world, constants.data(), n));
        if (n) mConstants = constants[n - 1];
    }

    // n frames = n x { update(...); render(...) } with the same bits in the output, through shared sets of launches
    // (rt_pipeline_render_batch; the sample-batch mode of long accumulations).  elapsedFrames is the frame count of the FIRST
    // of them; the host state (jitter RNG, accumulation counter) advances exactly as n update() calls would advance it.
    void renderBatch(float elapsedTime, UINT firstElapsedFrames, UINT n, UINT width, UINT height)
    {
        std::vector<PerFrameConstants> constants(n);
        float cam[11];
        mCamera->Pack(cam);
        rt_debug_options *opt = nullptr;
        DXRFramework::ThrowIfFailed(rt_progressive_host_options(mHost, &opt));
        *opt = mShaderDebugOptions;
        DXRFramework::ThrowIfFailed(rt_progressive_host_set_flags(mHost, mFrameAccumulationEnabled, mAnimationPaused));
        for (UINT i = 0; i < n; ++i)
            DXRFramework::ThrowIfFailed(rt_progressive_host_update(mHost, cam, elapsedTime, firstElapsedFrames + i, width, height, &constants[i]));
        fillShaderTable();
        DXRFramework::ThrowIfFailed(rt_pipeline_render_batch(mPipeline, width, height, constants.data(), n));
        if (n) mConstants = constants[n - 1];
    }

    // Sets of frames behind update() + render() (rt_pipeline_set_deferred).  OFF by default (round 5): render() renders,
    // as the reference's does (src/DXRExperimentsApp.cpp:194: DispatchRays, then the copy of the output, every frame), so a caller
    // that fetched getOutputResource() once and synchronises its own stream always sees the frame it asked for.  A headless
    // accumulation opts in with setDeferredFrames(n <= 32): render() then records the frame, and the recorded frames go through
    // one set of launches when n have gathered or when anything reads or changes what they produce -- getOutputResource() /
    // readOutput() / saveCheckpoint(), a material or scene change, RtContext::synchronize, rt_context_get_stream,
    // rt_device_download, the pipeline's destruction.  Bit for bit the image of rendering every frame at once.  0 or 1: off.
    void setDeferredFrames(UINT frames) { DXRFramework::ThrowIfFailed(rt_pipeline_set_deferred(mPipeline, frames)); }
    UINT getDeferredFrames() const { uint32_t m = 0; rt_pipeline_get_deferred(mPipeline, &m, nullptr); return m; }
    UINT getPendingFrames() const { uint32_t n = 0; rt_pipeline_get_deferred(mPipeline, nullptr, &n); return n; }
    void flush() { DXRFramework::ThrowIfFailed(rt_pipeline_flush(mPipeline)); }

    // the light buffer of occluders shadow rays test first (-1 automatic, 0 off, else cells per side): same image, less time
    void setShadowCache(int cellsPerSide) { DXRFramework::ThrowIfFailed(rt_pipeline_set_shadow_cache(mPipeline, cellsPerSide)); }

    // radius of the empty sphere around the point light that its shadow rays stop at (0: not known yet / none)
    float getFreeSphere() { float r = 0.0f; DXRFramework::ThrowIfFailed(rt_pipeline_get_free_sphere(mPipeline, &r)); return r; }

    // sizes the work memory of renderBatch calls of n frames ahead of time (optional)
    void reserveBatch(UINT n, UINT width, UINT height) { DXRFramework::ThrowIfFailed(rt_pipeline_reserve_batch(mPipeline, width, height, n)); }

    // what render() does before DispatchRays (:217-240): per-instance hit records, miss records, apply
    void fillShaderTable()
    {
        auto program = mRtBindings->getProgram();
        for (UINT rayType = 0; rayType < program->getHitProgramCount(); ++rayType) {
            for (UINT instance = 0; instance < mRtScene->getNumInstances(); ++instance) {
                auto &hitVars = mRtBindings->getHitVars(rayType, instance);
                hitVars->appendHeapRanges(0);         // vertex buffer SRV: geometry is owned by the library
                hitVars->appendHeapRanges(0);         // index buffer SRV
                const Material &m = mMaterials[instance < mMaterials.size() ? instance : mMaterials.size() - 1];
                hitVars->append32BitConstants(&m.params, sizeof(MaterialParams) / 4);
            }
        }
        for (UINT rayType = 0; rayType < program->getMissProgramCount(); ++rayType) {
            auto &missVars = mRtBindings->getMissVars(rayType);
            missVars->appendHeapRanges(0);            // envMap (bound but unused by the shaders)
            missVars->appendHeapRanges(0);            // envCubemap: set by loadResources / setEnvironment*
        }
        mRtBindings->apply(mRtContext, mRtState);
    }

    // loadResources (:104-125): the radiance cube map.  The reference's path is hard-coded; pass "" for a
    // constant grey environment.
    virtual void loadResources(UINT frameCount) override { (void)frameCount; }
    void loadEnvironmentDDS(const std::string &path) { DXRFramework::ThrowIfFailed(rt_pipeline_load_environment_dds(mPipeline, path.c_str())); }
    void setEnvironmentCube(const float *facesRGBA32F, uint32_t size) { DXRFramework::ThrowIfFailed(rt_pipeline_set_environment_cube(mPipeline, facesRGBA32F, size)); }
    void setEnvironmentConstant(float r, float g, float b) { const float c[3] = {r, g, b}; DXRFramework::ThrowIfFailed(rt_pipeline_set_environment_constant(mPipeline, c)); }

    virtual void createOutputResource(UINT format, UINT width, UINT height) override { DXRFramework::ThrowIfFailed(rt_pipeline_create_output(mPipeline, format, width, height)); mWidth = width; mHeight = height; mFormat = format; }
    // the reference's output texture is RGBA16F and RayGen read-modify-writes it every frame (src/DXRExperimentsApp.cpp:28,
    // assets/shaders/ProgressiveRaytracing.hlsl:36-38): setAccumulationStorage(RT_FORMAT_R16G16B16A16_FLOAT) rounds the running mean to
    // fp16 every frame as that texture does; the default keeps it fp32 (createOutputResource's format then only converts on read)
    void setAccumulationStorage(UINT format, UINT rounding = RT_ROUND_NEAREST_EVEN) { DXRFramework::ThrowIfFailed(rt_pipeline_set_accumulation_storage(mPipeline, format, rounding)); }
    virtual void buildAccelerationStructures() override
    {
        mRtScene->build(mRtContext, mRtProgram->getHitProgramCount());
    }

    virtual void addMaterial(Material material) override { mMaterials.push_back(material); }
    virtual void setCamera(std::shared_ptr<Math::Camera> camera) override { mCamera = camera; }
    virtual void setScene(DXRFramework::RtScene::SharedPtr scene) override
    {
        mRtScene = scene;
        mRtBindings = DXRFramework::RtBindings::create(mRtContext, mRtProgram, scene);
        mRtBindings->bindPipeline(mPipeline);
        DXRFramework::ThrowIfFailed(rt_pipeline_set_scene(mPipeline, scene->getHandle(mRtContext)));
    }

    virtual int getNumOutputs() override { return 1; }
    virtual void *getOutputResource(UINT id) override { void *p = nullptr; DXRFramework::ThrowIfFailed(rt_pipeline_get_output_device_ptr(mPipeline, id, &p)); return p; }
    // copy the accumulation image to the host (RGBA32F, or RGBA16F if created with that format)
    void readOutput(void *host, size_t bytes) { DXRFramework::ThrowIfFailed(rt_pipeline_read_output(mPipeline, host, bytes)); }
    // accumulation checkpoint: image + mAccumCount / last camera / options / RNG; resuming continues bit for bit
    void saveCheckpoint(const char *path) { DXRFramework::ThrowIfFailed(rt_pipeline_save_checkpoint(mPipeline, mHost, path)); }
    void loadCheckpoint(const char *path) { DXRFramework::ThrowIfFailed(rt_pipeline_load_checkpoint(mPipeline, mHost, path)); }

    virtual bool *isActive() override { return &mActive; }
    virtual const char *getName() override { return rt_pipeline_get_name(mPipeline); }

    DebugOptions &debugOptions() { return mShaderDebugOptions; }          // what the ImGui panel edits (:288-301)
    void setFrameAccumulation(bool on) { mFrameAccumulationEnabled = on; mAnimationPaused = true; }
    void setAnimationPaused(bool on) { mAnimationPaused = on; }
    void invalidateAccumulation() { rt_progressive_host_reset(mHost); }    // frameDirty (:309-311)
    const PerFrameConstants &constants() const { return mConstants; }
    rt_pipeline *getHandle() const { return mPipeline; }

private:
    ProgressiveRaytracingPipeline(DXRFramework::RtContext::SharedPtr context, uint32_t rngSeed) : mRtContext(context)
    {
        using namespace DXRFramework;
        RtProgram::Desc programDesc;
        programDesc.addShaderLibrary({L"RayGen", L"PrimaryClosestHit", L"PrimaryMiss", L"ShadowClosestHit", L"ShadowAnyHit", L"ShadowMiss"});
        programDesc.setRayGen("RayGen");
        programDesc.addHitGroup(0, "PrimaryClosestHit", "").addMiss(0, "PrimaryMiss");
        programDesc.addHitGroup(1, "ShadowClosestHit", "ShadowAnyHit").addMiss(1, "ShadowMiss");
        mRtProgram = RtProgram::create(context, programDesc);
        mRtState = RtState::create(context);
        mRtState->setProgram(mRtProgram);
        mRtState->setMaxTraceRecursionDepth(4);
        mRtState->setMaxAttributeSize(8);
        mRtState->setMaxPayloadSize(20);
        ThrowIfFailed(rt_pipeline_create(context->getHandle(), RT_PIPELINE_PROGRESSIVE, &mPipeline));
        ThrowIfFailed(rt_progressive_host_create(rngSeed, &mHost));
        rt_debug_options *opt = nullptr;
        ThrowIfFailed(rt_progressive_host_options(mHost, &opt));
        mShaderDebugOptions = *opt;                 // reference defaults (:74-84)
        std::memset(&mConstants, 0, sizeof mConstants);
    }

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
    DebugOptions mShaderDebugOptions;
    UINT mWidth = 0, mHeight = 0, mFormat = RT_FORMAT_R32G32B32A32_FLOAT;
    bool mActive = true, mFrameAccumulationEnabled = true, mAnimationPaused = true;
};
