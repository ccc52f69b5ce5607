// Headless stand-in for the reference app's second mode (src/DXRExperimentsApp.cpp:119-130, 194-211):
// the RealtimeRaytracingPipeline renders its two AOVs (direct lighting, indirect specular) and the
// DenoiseCompositor filters and tone-maps them into the displayed image, every frame.  Uses only the
// reference-shaped C++ API (dxrexperiments_amd/include); the denoised last frame is written as a PNG or PFM.
// (src/DXRExperimentsApp.cpp:135-137 creates the compositor, :194-211 feeds it the two outputs every frame.)
//
//   realtime_denoise <model.obj> <width> <height> <frames> <out.png|out.pfm> [eye.x eye.y eye.z at.x at.y at.z]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "DenoiseCompositor.h"
#include "RealtimeRaytracingPipeline.h"

using namespace DXRFramework;

int main(int argc, char **argv)
{
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s model.obj width height frames out.png|out.pfm [eye xyz at xyz]\n", argv[0]);
        return 2;
    }
    const UINT width = std::atoi(argv[2]), height = std::atoi(argv[3]), frames = std::atoi(argv[4]);
    try {
        auto context = RtContext::create(0);
        auto scene = RtScene::create();
        scene->addModel(RtModel::create(context, argv[1]), Matrix::identity());

        RaytracingPipeline::Material material{};                       // DXRExperimentsApp.cpp:95-104
        material.params.albedo = {0.95f, 0.05f, 0.0f, 1.0f};
        material.params.specular = {0.58f, 0.58f, 0.58f, 1.0f};
        material.params.roughness = 0.5f;
        material.params.reflectivity = 0.7f;
        material.params.type = 1;

        auto camera = std::make_shared<Math::Camera>();
        camera->SetAspectRatio(float(width) / float(height));
        if (argc >= 12)
            camera->SetEyeAtUp({(float)std::atof(argv[6]), (float)std::atof(argv[7]), (float)std::atof(argv[8])},
                               {(float)std::atof(argv[9]), (float)std::atof(argv[10]), (float)std::atof(argv[11])}, {0, 1, 0});
        else
            camera->SetEyeAtUp({0.0f, 0.0f, 3.2f}, {0.0f, 0.0f, 0.0f}, {0, 1, 0});

        auto pipeline = RealtimeRaytracingPipeline::create(context);
        pipeline->setScene(scene);
        pipeline->addMaterial(material);
        pipeline->setCamera(camera);
        pipeline->loadResources(3);
        pipeline->createOutputResource(RT_FORMAT_R32G32B32A32_FLOAT, width, height);
        pipeline->buildAccelerationStructures();

        auto denoiser = DenoiseCompositor::create(context);
        denoiser->loadResources(3, false);
        denoiser->createOutputResource(RT_FORMAT_R32G32B32A32_FLOAT, width, height);

        const auto t0 = std::chrono::steady_clock::now();
        for (UINT frame = 1; frame <= frames; ++frame) {
            pipeline->update(0.0f, frame, (frame + 2) % 3, frame % 3, width, height);
            pipeline->render(frame % 3, width, height);
            // src/DXRExperimentsApp.cpp:198-212, line for line (commandList: unused here)
            ID3D12GraphicsCommandList *commandList = nullptr;
            for (int i = 0; i < pipeline->getNumOutputs(); ++i)
                context->transitionResource(pipeline->getOutputResource(i), D3D12_RESOURCE_STATE_UNORDERED_ACCESS, D3D12_RESOURCE_STATE_NON_PIXEL_SHADER_RESOURCE);
            DenoiseCompositor::InputComponents inputs = {};
            inputs.directLightingSrv = pipeline->getOutputSrvHandle(0);
            inputs.indirectSpecularSrv = pipeline->getOutputSrvHandle(1);
            denoiser->dispatch(commandList, inputs, frame % 3, width, height);
            for (int i = 0; i < pipeline->getNumOutputs(); ++i)
                context->transitionResource(pipeline->getOutputResource(i), D3D12_RESOURCE_STATE_NON_PIXEL_SHADER_RESOURCE, D3D12_RESOURCE_STATE_UNORDERED_ACCESS);
        }
        std::vector<float> image(size_t(width) * height * 4);
        denoiser->readOutput(image.data(), image.size() * sizeof(float));
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%s + denoise: %u frames, %.2f fps, ~%.2f Million Primary Rays/s\n", pipeline->getName(), frames, frames / s,
                    double(width) * height * frames / s / 1e6);

        // the compositor already tone-mapped and gamma-corrected: the PNG is a plain 8-bit quantisation
        const std::string out = argv[5];
        const bool png = out.size() > 4 && out.compare(out.size() - 4, 4, ".png") == 0;
        ThrowIfFailed(png ? rt_image_write_png(out.c_str(), image.data(), width, height, 1.0f, 1.0f, 0)
                          : rt_image_write_pfm(out.c_str(), image.data(), width, height));
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
