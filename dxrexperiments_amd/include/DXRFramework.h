// DXRFramework.h -- header-only C++ mirror of the reference's DXRFramework wrapper
// (libs/DXRFramework/Rt{Context,Model,Scene,Program,State,Bindings,Params}.h) over the
// C ABI in include/dxr_amd.h.  Same class names, same factory / method names and argument
// meaning, same error behaviour (exceptions), so host code written against the reference
// keeps its shape; D3D12 handle types are replaced as listed below.
//
//   ID3D12Device* / ID3D12GraphicsCommandList*   -> device ordinal (+ optional hipStream_t)
//   ID3D12Resource* / descriptor handles          -> device pointers owned by the library
//   DXIL libraries / shader identifiers           -> names of the built-in entry points;
//                                                    RtProgram::Desc records and validates them,
//                                                    there is no runtime shader linking on HIP
//   shader-table records (RtBindings / RtParams)  -> root constants of the hit records are
//                                                    forwarded as per-instance materials
//   DirectX::XMMATRIX                             -> DXRFramework::Matrix (row-vector 4x4, row major)
#pragma once

#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "dxr_amd.h"

// D3D12 types the reference's call sites name, kept so that they compile unchanged (src/DXRExperimentsApp.cpp:186-215):
//   D3D12_GPU_DESCRIPTOR_HANDLE   the reference hands SRV / UAV descriptor-heap handles of its output textures to the
//                                 denoiser; here the handle's 64-bit `ptr` is the DEVICE ADDRESS of the RGBA32F image
//   ID3D12GraphicsCommandList     opaque and never defined: work goes to the RtContext's stream, the argument is ignored
//   D3D12_RESOURCE_STATES         resource barriers have no HIP analogue on one in-order stream: transitionResource is a no-op
struct D3D12_GPU_DESCRIPTOR_HANDLE { unsigned long long ptr; };
struct ID3D12GraphicsCommandList;
enum D3D12_RESOURCE_STATES { D3D12_RESOURCE_STATE_UNORDERED_ACCESS = 0x8, D3D12_RESOURCE_STATE_NON_PIXEL_SHADER_RESOURCE = 0x40 };

namespace DXRFramework
{
    // what the reference's ThrowIfFailed / HrException do (Helpers/DirectXHelper.h:22-64)
    class RtException : public std::runtime_error
    {
    public:
        RtException(int code, const std::string &what) : std::runtime_error(what), mCode(code) {}
        int code() const { return mCode; }
    private:
        int mCode;
    };

    inline void ThrowIfFailed(int rc)
    {
        if (rc != RT_OK) throw RtException(rc, rt_last_error());
    }

    // Row-vector convention like DirectX::XMMATRIX: v' = v * M, translation in row 3.
    struct Matrix
    {
        float m[4][4];
        static Matrix identity()
        {
            Matrix r;
            std::memset(r.m, 0, sizeof r.m);
            r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0f;
            return r;
        }
        static Matrix translation(float x, float y, float z)
        {
            Matrix r = identity();
            r.m[3][0] = x; r.m[3][1] = y; r.m[3][2] = z;
            return r;
        }
        // first three rows of the transpose: what D3D12_RAYTRACING_INSTANCE_DESC::Transform stores
        // (Helpers/TopLevelASGenerator.cpp:355-357)
        void toInstanceTransform(float out[12]) const
        {
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 4; ++c) out[4 * r + c] = m[c][r];
        }
    };

    class RtBindings;
    class RtState;

    // ---- RtContext (RtContext.h:11-58) -------------------------------------------------
    class RtContext
    {
    public:
        using SharedPtr = std::shared_ptr<RtContext>;

        // reference: create(ID3D12Device*, ID3D12GraphicsCommandList*, bool forceComputeFallback)
        static SharedPtr create(int device = 0, void *hipStream = nullptr, bool useCallerStream = false)
        {
            rt_context *h = nullptr;
            ThrowIfFailed(useCallerStream ? rt_context_create_on_stream(device, hipStream, &h) : rt_context_create(device, &h));
            return SharedPtr(new RtContext(h));
        }
        ~RtContext() { rt_context_destroy(mHandle); }

        rt_context *getHandle() const { return mHandle; }
        int getDevice() const { int d = 0; ThrowIfFailed(rt_context_get_device(mHandle, &d)); return d; }
        void *getStream() const { void *s = nullptr; ThrowIfFailed(rt_context_get_stream(mHandle, &s)); return s; }
        bool isUsingNativeDxr() const { return false; }
        void waitForGpu() { ThrowIfFailed(rt_context_synchronize(mHandle)); }
        // RtContext::transitionResource (RtContext.cpp:224-232): kernels of one stream run in order, nothing to do
        void transitionResource(void *, D3D12_RESOURCE_STATES, D3D12_RESOURCE_STATES) {}

        // raytrace(bindings, state, width, height, depth) (RtContext.cpp:192-222); defined below
        inline void raytrace(std::shared_ptr<RtBindings> bindings, std::shared_ptr<RtState> state, uint32_t width, uint32_t height, uint32_t depth);

    private:
        explicit RtContext(rt_context *h) : mHandle(h) {}
        rt_context *mHandle;
    };

    // ---- RtModel (RtModel.h:9-40) ------------------------------------------------------
    class RtModel
    {
    public:
        using SharedPtr = std::shared_ptr<RtModel>;

        static SharedPtr create(RtContext::SharedPtr context, const std::string &filePath)
        {
            rt_model *h = nullptr;
            int rc = rt_model_create_from_file(context->getHandle(), filePath.c_str(), &h);      // .fbx or .obj
            if (rc != RT_OK) {
                // the reference substitutes one triangle when the import fails (RtModel.cpp:58-68)
                static const rt_vertex tri[3] = {{{0.0f, 0.25f, 0.0f}, {0, 0, 1}}, {{0.25f, -0.25f, 0.0f}, {0, 0, 1}}, {{-0.25f, -0.25f, 0.0f}, {0, 0, 1}}};
                static const uint32_t idx[3] = {0, 2, 1};
                ThrowIfFailed(rt_model_create_from_arrays(context->getHandle(), tri, 3, idx, 1, &h));
            }
            return SharedPtr(new RtModel(context, h));
        }
        static SharedPtr create(RtContext::SharedPtr context, const rt_vertex *verts, uint32_t numVertices, const uint32_t *indices, uint32_t numTriangles)
        {
            rt_model *h = nullptr;
            ThrowIfFailed(rt_model_create_from_arrays(context->getHandle(), verts, numVertices, indices, numTriangles, &h));
            return SharedPtr(new RtModel(context, h));
        }
        ~RtModel() { rt_model_destroy(mHandle); }

        rt_model *getHandle() const { return mHandle; }
        uint32_t getNumVertices() const { uint32_t v = 0, t = 0; ThrowIfFailed(rt_model_get_counts(mHandle, &v, &t)); return v; }
        uint32_t getNumTriangles() const { uint32_t v = 0, t = 0; ThrowIfFailed(rt_model_get_counts(mHandle, &v, &t)); return t; }

    private:
        RtModel(RtContext::SharedPtr ctx, rt_model *h) : mContext(ctx), mHandle(h) {}
        RtContext::SharedPtr mContext;
        rt_model *mHandle;
    };

    // ---- RtScene (RtScene.h:9-49) ------------------------------------------------------
    class RtScene
    {
    public:
        using SharedPtr = std::shared_ptr<RtScene>;

        // the reference creates scenes before binding them to a context; the handle is made on first use
        static SharedPtr create() { return SharedPtr(new RtScene()); }
        ~RtScene() { if (mHandle) rt_scene_destroy(mHandle); }

        void addModel(RtModel::SharedPtr model, const Matrix &transform) { mInstances.push_back({model, transform}); }
        RtModel::SharedPtr getModel(uint32_t index) const { return mInstances[index].model; }
        uint32_t getNumInstances() const { return static_cast<uint32_t>(mInstances.size()); }

        // build(context, hitGroupCount) (RtScene.cpp:18-52): BLAS per model, then the TLAS
        void build(RtContext::SharedPtr context, uint32_t hitGroupCount)
        {
            realize(context);
            ThrowIfFailed(rt_scene_build(mHandle, hitGroupCount));
        }
        rt_scene *getHandle(RtContext::SharedPtr context) { realize(context); return mHandle; }
        float getBuildMilliseconds() const { float ms = 0; if (mHandle) rt_scene_build_ms(mHandle, &ms); return ms; }

    private:
        RtScene() = default;
        struct Node { RtModel::SharedPtr model; Matrix transform; };
        // ONE rt_scene handle per RtScene for its whole life (a pipeline that was given the handle by setScene keeps
        // seeing the scene the caller keeps editing, as with the reference's shared RtScene object): instances added
        // since the last call are appended to it
        void realize(RtContext::SharedPtr context)
        {
            if (!mHandle) {
                ThrowIfFailed(rt_scene_create(context->getHandle(), &mHandle));
                mContext = context;
            }
            for (; mRealized < mInstances.size(); mRealized++) {
                float x[12];
                mInstances[mRealized].transform.toInstanceTransform(x);
                ThrowIfFailed(rt_scene_add_model(mHandle, mInstances[mRealized].model->getHandle(), x));
            }
        }
        std::vector<Node> mInstances;
        RtContext::SharedPtr mContext;
        rt_scene *mHandle = nullptr;
        size_t mRealized = 0;
    };

    // ---- RtProgram (RtProgram.h:17-123): a description of which built-in entry points are used ----
    class RootSignatureGenerator {};   // placeholder for the D3D12 root-signature configurators

    class RtProgram
    {
    public:
        using SharedPtr = std::shared_ptr<RtProgram>;

        class Desc
        {
        public:
            Desc() = default;
            // reference: addShaderLibrary(bytecode, size, exports); here the "library" is the set of kernels
            // compiled into the HIP library, the exports are checked against it at create()
            Desc &addShaderLibrary(const std::vector<std::wstring> &symbolExports) { mExports.insert(mExports.end(), symbolExports.begin(), symbolExports.end()); return *this; }
            Desc &setRayGen(const std::string &raygen) { mRayGen = raygen; return *this; }
            Desc &addMiss(uint32_t missIndex, const std::string &miss)
            {
                if (mMiss.size() <= missIndex) mMiss.resize(missIndex + 1);
                mMiss[missIndex] = miss;
                return *this;
            }
            Desc &addHitGroup(uint32_t hitIndex, const std::string &closestHit, const std::string &anyHit, const std::string &intersection = "")
            {
                if (mHit.size() <= hitIndex) mHit.resize(hitIndex + 1);
                mHit[hitIndex] = {closestHit, anyHit, intersection};
                return *this;
            }
            using RootSignatureConfigurator = std::function<void(RootSignatureGenerator &config)>;
            Desc &configureGlobalRootSignature(RootSignatureConfigurator) { return *this; }
            Desc &configureRayGenRootSignature(RootSignatureConfigurator) { return *this; }
            Desc &configureHitGroupRootSignature(RootSignatureConfigurator) { return *this; }
            Desc &configureMissRootSignature(RootSignatureConfigurator) { return *this; }
        private:
            friend class RtProgram;
            struct Hit { std::string closestHit, anyHit, intersection; };
            std::string mRayGen;
            std::vector<std::string> mMiss;
            std::vector<Hit> mHit;
            std::vector<std::wstring> mExports;
        };

        static SharedPtr create(RtContext::SharedPtr context, const Desc &desc, uint32_t maxPayloadSize = 14 * sizeof(float), uint32_t maxAttributesSize = 32)
        {
            (void)maxPayloadSize; (void)maxAttributesSize;
            // the progressive kernels implement exactly this program (src/ProgressiveRaytracingPipeline.cpp:33-39)
            static const char *known[] = {"RayGen", "PrimaryClosestHit", "PrimaryMiss", "ShadowClosestHit", "ShadowAnyHit", "ShadowMiss", ""};
            auto check = [](const std::string &n) {
                for (const char *k : known) if (n == k) return;
                throw std::logic_error("RtProgram: entry point '" + n + "' is not built into the HIP library");
            };
            check(desc.mRayGen);
            for (const std::string &m : desc.mMiss) check(m);
            for (const Desc::Hit &h : desc.mHit) { check(h.closestHit); check(h.anyHit); check(h.intersection); }
            return SharedPtr(new RtProgram(context, desc));
        }
        uint32_t getHitProgramCount() const { return static_cast<uint32_t>(mDesc.mHit.size()); }
        uint32_t getMissProgramCount() const { return static_cast<uint32_t>(mDesc.mMiss.size()); }
        const std::string &getRayGenProgram() const { return mDesc.mRayGen; }

    private:
        RtProgram(RtContext::SharedPtr ctx, const Desc &d) : mContext(ctx), mDesc(d) {}
        RtContext::SharedPtr mContext;
        Desc mDesc;
    };

    // ---- RtState (RtState.h:10-43) -----------------------------------------------------
    class RtState
    {
    public:
        using SharedPtr = std::shared_ptr<RtState>;
        static SharedPtr create(RtContext::SharedPtr context) { return SharedPtr(new RtState(context)); }

        void setProgram(RtProgram::SharedPtr pProg) { mProgram = pProg; }
        RtProgram::SharedPtr getProgram() const { return mProgram; }
        void setMaxTraceRecursionDepth(uint32_t maxDepth) { mMaxTraceRecursionDepth = maxDepth; }
        uint32_t getMaxTraceRecursionDepth() const { return mMaxTraceRecursionDepth; }
        void setMaxPayloadSize(uint32_t maxSize) { mMaxPayloadSize = maxSize; }
        uint32_t getMaxPayloadSize() const { return mMaxPayloadSize; }
        void setMaxAttributeSize(uint32_t maxSize) { mMaxAttributeSize = maxSize; }
        uint32_t getMaxAttributeSize() const { return mMaxAttributeSize; }

    private:
        explicit RtState(RtContext::SharedPtr ctx) : mContext(ctx) {}
        RtContext::SharedPtr mContext;
        RtProgram::SharedPtr mProgram;
        uint32_t mMaxTraceRecursionDepth = 1, mMaxPayloadSize = 20, mMaxAttributeSize = 8;
    };

    // ---- RtParams (RtParams.h:9-52): root arguments of one shader record ----------------
    class RtParams
    {
    public:
        using SharedPtr = std::shared_ptr<RtParams>;
        static SharedPtr create(uint32_t initialOffset = 0) { return SharedPtr(new RtParams(initialOffset)); }

        void allocateStorage(uint32_t sizeInBytes) { mCapacity = sizeInBytes; }
        void appendHeapRanges(uint64_t gpuHandle) { push(&gpuHandle, 8); }
        void appendDescriptor(uint64_t descriptorHandle) { push(&descriptorHandle, 8); }
        void append32BitConstants(const void *constants, uint32_t num32BitConstants)
        {
            mConstantsOffset = mData.size();
            mNumConstants = num32BitConstants;
            push(constants, 4u * num32BitConstants);
        }
        // storage is consumed by every apply(), callers re-append each frame (RtParams.cpp:17-27)
        void reset() { mData.clear(); mNumConstants = 0; }
        const void *constants() const { return mNumConstants ? mData.data() + mConstantsOffset : nullptr; }
        uint32_t numConstants() const { return mNumConstants; }

    private:
        explicit RtParams(uint32_t initialOffset) : mInitialOffset(initialOffset) {}
        void push(const void *p, size_t n)
        {
            if (mData.size() + n > mCapacity) return;      // out-of-bounds writes are dropped with a log line in the reference (RtParams.cpp:35-37)
            const uint8_t *b = static_cast<const uint8_t *>(p);
            mData.insert(mData.end(), b, b + n);
        }
        std::vector<uint8_t> mData;
        size_t mCapacity = 80, mConstantsOffset = 0;
        uint32_t mNumConstants = 0, mInitialOffset;
    };

    // ---- RtBindings (RtBindings.h:12-67): the shader table ------------------------------
    class RtBindings
    {
    public:
        using SharedPtr = std::shared_ptr<RtBindings>;
        static SharedPtr create(RtContext::SharedPtr context, RtProgram::SharedPtr program, RtScene::SharedPtr scene)
        {
            return SharedPtr(new RtBindings(context, program, scene));
        }

        // (the table grows when instances were added to the scene after the bindings were made)
        const RtParams::SharedPtr &getHitVars(uint32_t rayID, uint32_t meshID)
        {
            auto &perRay = mHitParams.at(rayID);
            while (perRay.size() <= meshID) perRay.push_back(RtParams::create(32));
            return perRay[meshID];
        }
        const RtParams::SharedPtr &getRayGenVars() { return mRayGenParams; }
        const RtParams::SharedPtr &getMissVars(uint32_t rayID) { return mMissParams[rayID]; }
        const RtProgram::SharedPtr &getProgram() { return mProgram; }
        uint32_t getHitProgramsCount() const { return mProgram->getHitProgramCount(); }
        uint32_t getMissProgramsCount() const { return mProgram->getMissProgramCount(); }
        uint32_t getRecordSize() const { return 128; }      // ROUND_UP(32 + 80, 32) (RtBindings.cpp:60-61)

        // apply(context, state) (RtBindings.cpp:100-129): write every record.  Here: the 16 root constants of
        // the ray-type-0 hit record of instance i become material i of the bound pipeline.
        void apply(RtContext::SharedPtr, std::shared_ptr<RtState>)
        {
            if (!mPipeline) throw std::logic_error("RtBindings::apply: no pipeline bound");
            for (uint32_t inst = 0; inst < mHitParams[0].size(); ++inst) {
                RtParams::SharedPtr p = mHitParams[0][inst];
                if (p->numConstants() == sizeof(rt_material_params) / 4) {
                    rt_material_params m;
                    std::memcpy(&m, p->constants(), sizeof m);
                    if (inst < mMaterialsSet) ThrowIfFailed(rt_pipeline_set_material(mPipeline, inst, &m));
                    else { ThrowIfFailed(rt_pipeline_add_material(mPipeline, &m)); mMaterialsSet++; }
                }
            }
            for (auto &perRay : mHitParams) for (auto &p : perRay) p->reset();
            for (auto &p : mMissParams) p->reset();
        }
        void bindPipeline(rt_pipeline *p) { mPipeline = p; }
        rt_pipeline *getPipeline() const { return mPipeline; }

    private:
        RtBindings(RtContext::SharedPtr, RtProgram::SharedPtr program, RtScene::SharedPtr scene) : mProgram(program), mScene(scene)
        {
            mRayGenParams = RtParams::create(32);
            mHitParams.resize(program->getHitProgramCount());
            for (auto &perRay : mHitParams)
                for (uint32_t j = 0; j < scene->getNumInstances(); ++j) perRay.push_back(RtParams::create(32));
            for (uint32_t i = 0; i < program->getMissProgramCount(); ++i) mMissParams.push_back(RtParams::create(32));
        }
        RtProgram::SharedPtr mProgram;
        RtScene::SharedPtr mScene;
        RtParams::SharedPtr mRayGenParams;
        std::vector<std::vector<RtParams::SharedPtr>> mHitParams;
        std::vector<RtParams::SharedPtr> mMissParams;
        rt_pipeline *mPipeline = nullptr;
        uint32_t mMaterialsSet = 0;
    };

    inline void RtContext::raytrace(std::shared_ptr<RtBindings> bindings, std::shared_ptr<RtState>, uint32_t width, uint32_t height, uint32_t depth)
    {
        (void)depth;      // the reference passes Depth = 3 and only uses .xy of the launch index (ProgressiveRaytracingPipeline.cpp:244)
        ThrowIfFailed(rt_pipeline_render(bindings->getPipeline(), width, height));
    }
}
