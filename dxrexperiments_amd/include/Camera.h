// Camera.h -- the slice of MiniEngine's Math::Camera (libs/MiniEngine/Camera.h:20-173) the
// pipelines consume: eye / look-at / up, vertical FOV and aspect ratio.  The path reads
// four vectors only (eye, U, V, W; ProgressiveRaytracingPipeline.cpp:151-168); input handling,
// frustums and projection matrices of MiniEngine are out of scope.
#pragma once

#include <cmath>
#include <cstring>

namespace Math
{
    struct Vector3
    {
        float x, y, z;
        Vector3(float x_ = 0, float y_ = 0, float z_ = 0) : x(x_), y(y_), z(z_) {}
    };

    class Camera
    {
    public:
        Camera() : mEye(0, 0, 0), mAt(0, 0, -1), mUp(0, 1, 0) {}

        void SetEyeAtUp(Vector3 eye, Vector3 at, Vector3 up) { mEye = eye; mAt = at; mUp = up; }     // Camera.h:125-129
        void SetPosition(Vector3 p) { Vector3 d(mAt.x - mEye.x, mAt.y - mEye.y, mAt.z - mEye.z); mEye = p; mAt = Vector3(p.x + d.x, p.y + d.y, p.z + d.z); }
        void SetFOV(float verticalFovInRadians) { mVerticalFOV = verticalFovInRadians; }
        void SetAspectRatio(float widthOverHeight) { mAspectRatio = widthOverHeight; }                 // the app passes width/height (DXSample.cpp:44)
        void SetZRange(float nearZ, float farZ) { mNear = nearZ; mFar = farZ; }
        void Update() {}

        Vector3 GetPosition() const { return mEye; }
        Vector3 GetLookAt() const { return mAt; }
        Vector3 GetUpHint() const { return mUp; }
        float GetFOV() const { return mVerticalFOV; }
        float GetAspectRatio() const { return mAspectRatio; }

        // eye[3] at[3] up[3] fov aspect: the array rt_progressive_host_update takes
        void Pack(float out[11]) const
        {
            const float v[11] = {mEye.x, mEye.y, mEye.z, mAt.x, mAt.y, mAt.z, mUp.x, mUp.y, mUp.z, mVerticalFOV, mAspectRatio};
            std::memcpy(out, v, sizeof v);
        }

    private:
        Vector3 mEye, mAt, mUp;
        float mVerticalFOV = 3.14159265358979f / 4.0f;      // XM_PIDIV4 (Camera.h:143)
        float mAspectRatio = 16.0f / 9.0f;
        float mNear = 1.0f, mFar = 1000.0f;
    };
}
