// Drives the reference-shaped C++ API end to end and dumps the raw fp32 accumulation image so that the
// Python test can compare it with the committed golden frames (tests/golden/cornell64_golden.npz).
//   test_wrapper <cornell.obj> <out.f32>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

#include "DenoiseCompositor.h"
#include "ProgressiveRaytracingPipeline.h"
#include "RealtimeRaytracingPipeline.h"

using namespace DXRFramework;

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    try {
        const UINT W = 64, H = 64;
        auto context = RtContext::create(0);
        auto scene = RtScene::create();
        scene->addModel(RtModel::create(context, argv[1]), Matrix::identity());
        RaytracingPipeline::Material material{};
        material.params.albedo = {0.95f, 0.05f, 0.0f, 1.0f};
        material.params.specular = {0.58f, 0.58f, 0.58f, 1.0f};
        material.params.roughness = 0.5f;
        material.params.reflectivity = 0.7f;
        material.params.type = 1;
        auto camera = std::make_shared<Math::Camera>();
        camera->SetAspectRatio(1.0f);
        camera->SetEyeAtUp({0.0f, 0.0f, 3.2f}, {0.0f, 0.0f, 0.0f}, {0, 1, 0});
        RaytracingPipeline::SharedPtr pipeline = ProgressiveRaytracingPipeline::create(context, 1234);
        pipeline->setScene(scene);
        pipeline->addMaterial(material);
        pipeline->setCamera(camera);
        pipeline->loadResources(3);
        pipeline->createOutputResource(RT_FORMAT_R32G32B32A32_FLOAT, W, H);
        pipeline->buildAccelerationStructures();
        if (pipeline->getNumOutputs() != 1 || std::string(pipeline->getName()) != "Progressive Ray Tracing Pipeline") return 3;
        // (round 5) render() renders unless the caller opts in: a fresh pipeline holds no frames back
        if (static_cast<ProgressiveRaytracingPipeline *>(pipeline.get())->getDeferredFrames() != 0) return 18;
        static_cast<ProgressiveRaytracingPipeline *>(pipeline.get())->setDeferredFrames(32);
        for (UINT frame = 1; frame <= 4; ++frame) {
            pipeline->update(0.0f, frame, 0, 0, W, H);
            pipeline->render(0, W, H);
        }
        // (round 4) opted in, the mirror records the frames and renders them in sets behind update() + render(): four frames are
        // held until something reads them
        if (static_cast<ProgressiveRaytracingPipeline *>(pipeline.get())->getDeferredFrames() != 32 ||
            static_cast<ProgressiveRaytracingPipeline *>(pipeline.get())->getPendingFrames() != 4) return 14;
        std::vector<float> image(size_t(W) * H * 4);
        static_cast<ProgressiveRaytracingPipeline *>(pipeline.get())->readOutput(image.data(), image.size() * 4);
        if (static_cast<ProgressiveRaytracingPipeline *>(pipeline.get())->getPendingFrames() != 0) return 15;
        // ... and the image is the one of rendering every frame at once
        {
            auto immediate = ProgressiveRaytracingPipeline::create(context, 1234);
            immediate->setDeferredFrames(0);
            immediate->setScene(scene);
            immediate->addMaterial(material);
            immediate->setCamera(camera);
            immediate->loadResources(3);
            immediate->createOutputResource(RT_FORMAT_R32G32B32A32_FLOAT, W, H);
            immediate->buildAccelerationStructures();
            for (UINT frame = 1; frame <= 4; ++frame) {
                immediate->update(0.0f, frame, 0, 0, W, H);
                immediate->render(0, W, H);
                if (immediate->getPendingFrames() != 0) return 16;
            }
            std::vector<float> again(image.size());
            immediate->readOutput(again.data(), again.size() * 4);
            if (std::memcmp(again.data(), image.data(), image.size() * 4) != 0) return 17;
        }
        FILE *f = std::fopen(argv[2], "wb");
        if (!f) return 4;
        std::fwrite(image.data(), 4, image.size(), f);
        std::fclose(f);
        // the same four frames through ONE set of launches (renderBatch -> rt_pipeline_render_batch): the same bits
        {
            auto batched = ProgressiveRaytracingPipeline::create(context, 1234);
            batched->setScene(scene);
            batched->addMaterial(material);
            batched->setCamera(camera);
            batched->loadResources(3);
            batched->createOutputResource(RT_FORMAT_R32G32B32A32_FLOAT, W, H);
            batched->buildAccelerationStructures();
            batched->renderBatch(0.0f, 1, 4, W, H);
            std::vector<float> again(image.size());
            batched->readOutput(again.data(), again.size() * 4);
            if (std::memcmp(again.data(), image.data(), image.size() * 4) != 0) return 13;
        }
        // the reference's accumulation storage (an RGBA16F texture read-modify-written every frame, src/DXRExperimentsApp.cpp:28): the mirror's
        // setAccumulationStorage rounds the running mean to fp16 every frame -- every stored value is then an fp16 number, and the image differs
        // from the fp32 accumulation by rounding only
        {
            auto half = ProgressiveRaytracingPipeline::create(context, 1234);
            half->setDeferredFrames(0);
            half->setScene(scene);
            half->addMaterial(material);
            half->setCamera(camera);
            half->loadResources(3);
            half->createOutputResource(RT_FORMAT_R32G32B32A32_FLOAT, W, H);
            half->buildAccelerationStructures();
            half->setAccumulationStorage(RT_FORMAT_R16G16B16A16_FLOAT);
            for (UINT frame = 1; frame <= 4; ++frame) {
                half->update(0.0f, frame, 0, 0, W, H);
                half->render(0, W, H);
            }
            std::vector<float> h16(image.size());
            half->readOutput(h16.data(), h16.size() * 4);
            bool differs = false;
            for (size_t k = 0; k < h16.size(); k++) {
                uint32_t u;
                std::memcpy(&u, &h16[k], 4);
                if ((u & 0x1FFFu) != 0 && h16[k] >= 6.103515625e-05f) return 18;      // a normal fp16 number has 13 zero mantissa bits in fp32
                const float d = h16[k] - image[k];
                if (d != 0.0f) differs = true;
                if (d > 2e-3f * (image[k] > 1.0f ? image[k] : 1.0f) || -d > 2e-3f * (image[k] > 1.0f ? image[k] : 1.0f)) return 19;
            }
            if (!differs) return 20;
        }
        // error behaviour: a missing model falls back to the reference's single triangle, bad programs throw
        auto fallback = RtModel::create(context, "/nonexistent.obj");
        if (fallback->getNumTriangles() != 1) return 5;
        bool threw = false;
        try { RtProgram::Desc d; d.setRayGen("NoSuchShader"); RtProgram::create(context, d); } catch (const std::logic_error &) { threw = true; }
        if (!threw) return 6;
        // the app's second pipeline + post chain (src/DXRExperimentsApp.cpp:196-211): realtime AOVs -> denoiser
        if (argc >= 4) {
            auto realtime = RealtimeRaytracingPipeline::create(context, 77);
            realtime->setScene(scene);
            realtime->addMaterial(material);
            realtime->setCamera(camera);
            realtime->loadResources(3);
            realtime->createOutputResource(RT_FORMAT_R32G32B32A32_FLOAT, W, H);
            realtime->buildAccelerationStructures();
            if (realtime->getNumOutputs() != 2 || std::string(realtime->getName()) != "Realtime Ray Tracing Pipeline") return 7;
            realtime->update(0.0f, 5, 0, 0, W, H);
            realtime->render((ID3D12GraphicsCommandList *)nullptr, 0, W, H);      // the reference's signature
            auto denoiser = DenoiseCompositor::create(context);
            denoiser->loadResources(3, false);
            denoiser->createOutputResource(RT_FORMAT_R32G32B32A32_FLOAT, W, H);
            // the reference's call site (src/DXRExperimentsApp.cpp:202-206): SRV handles from the pipeline, command list first
            DenoiseCompositor::InputComponents inputs = {};
            inputs.directLightingSrv = realtime->getOutputSrvHandle(0);
            inputs.indirectSpecularSrv = realtime->getOutputSrvHandle(1);
            if (inputs.directLightingSrv.ptr != (unsigned long long)(size_t)realtime->getOutputResource(0) ||
                realtime->getOutputUavHandle(1).ptr != (unsigned long long)(size_t)realtime->getOutputResource(1)) return 12;
            denoiser->dispatch((ID3D12GraphicsCommandList *)nullptr, inputs, 0, W, H);
            std::vector<float> three(size_t(W) * H * 4 * 3);
            realtime->readOutput(0, three.data(), image.size() * 4);
            realtime->readOutput(1, three.data() + image.size(), image.size() * 4);
            denoiser->readOutput(three.data() + 2 * image.size(), image.size() * 4);
            FILE *g = std::fopen(argv[3], "wb");
            if (!g) return 8;
            std::fwrite(three.data(), 4, three.size(), g);
            std::fclose(g);
        }
        // addModel AFTER setScene: the pipeline holds the same scene handle, so a rebuild through the
        // pipeline renders the new instance too (the reference builds through the RtScene object the pipeline holds)
        {
            scene->addModel(RtModel::create(context, argv[1]), Matrix::translation(0.6f, 0.0f, 0.0f));
            if (scene->getNumInstances() != 2) return 9;
            pipeline->addMaterial(material);
            pipeline->buildAccelerationStructures();
            pipeline->update(0.0f, 9, 0, 0, W, H);
            pipeline->render(0, W, H);
            std::vector<float> again(image.size());
            static_cast<ProgressiveRaytracingPipeline *>(pipeline.get())->readOutput(again.data(), again.size() * 4);
            if (again == image) return 10;
        }
        std::printf("wrapper ok\n");
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
