// RtHip.cs — the C#/OpenTK host's binding to librt_hip.so (include/rt_hip.h, API version 7).
//
// Add this file to INFOGR2024Template.csproj (it needs <AllowUnsafeBlocks>true</AllowUnsafeBlocks>) and apply
// MyApplication.patch / objects.patch / template.patch from this directory.  Nothing else in the host changes:
// window, input, camera, scene construction, Surface, the TexImage2D upload and the FXAA present shader stay
// as they are.  Not compiled in the build image (no dotnet there); tests/test_abi_symbols.py checks every
// DllImport below against the library's exports and their parameter counts.
using System;
using System.Collections.Generic;
using System.Runtime.InteropServices;
using Objects;
using OpenTK.Mathematics;

namespace Template
{
    [StructLayout(LayoutKind.Sequential)]
    struct RtCamera            // rt_camera (40 bytes)
    {
        public float px, py, pz;      // Camera.position   (MyApplication.cs:261)
        public float dx, dy, dz;      // Camera.direction  (:262)
        public float ux, uy, uz;      // Camera.up         (:263)
        public float fovDeg;          // Camera.fov        (:264)
    }

    [StructLayout(LayoutKind.Sequential)]
    unsafe struct RtScene             // rt_scene (136 bytes)
    {
        public int numPrims, numLights;
        public int* primTypes;                          // 0 sphere / 1 plane / 2 triangle (MyApplication.cs:161,166,173)
        public float* primPositions, primNormals;       // Vector3[] = 3 consecutive floats per element
        public float* primDiffuse, primGlossy, primAmbient;   // Vector4[]
        public float* primSpecular, primRadius, primD;
        public float* v0, v1, v2;
        public int* primCheckerboard;                   // the host's bool[] widened to int[]
        public int* primTexture;                        // texture slot or -1 (Plane.texture, objects.cs:146)
        public float* lightPositions, lightColors;
    }

    [StructLayout(LayoutKind.Sequential)]
    struct RtStats             // rt_stats (160 bytes)
    {
        public ulong raysPrimary, raysShadow, raysSecondary, shadedHits;
        public ulong sphereTests, planeTests, triangleTests;          // Intersect calls of the reference's algorithm
        public ulong exactCandidates, filterRayTests, filterOriginTerms;
        public ulong primaryFilterRayTests, primaryExactCandidates, shadowCandidates;   // what the kernels executed
        public double kernelMs, primaryMs, traceMs, shadeMs, gatherMs, copyMs;
        public int numDevices, reserved;
    }

    [StructLayout(LayoutKind.Sequential)]
    struct RtDebugRay          // rt_debug_ray (56 bytes)
    {
        public Vector3 direction; public int prim;      // prim = index into scene.primitives, -1 = miss
        public Vector3 hit; public float t;
        public Vector3 normal, reflected;
    }

    static unsafe class RtHip
    {
        const string Lib = "rt_hip";                    // librt_hip.so
        public const int ReferenceMaxDepth = 5;         // the literal of `depth < 5`, objects.cs:84

        [DllImport(Lib)] public static extern int rt_api_version();
        [DllImport(Lib)] public static extern int rt_device_count();
        [DllImport(Lib)] public static extern int rt_create(out IntPtr ctx, int[] deviceIds, int nDevices);
        [DllImport(Lib)] public static extern void rt_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern IntPtr rt_last_error(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_set_scene(IntPtr ctx, ref RtScene scene);
        [DllImport(Lib)] public static extern int rt_set_texture(IntPtr ctx, int slot, byte[] rgba8, int width, int height);
        [DllImport(Lib)] public static extern int rt_render(IntPtr ctx, ref RtCamera camera, int width, int height, int maxDepth, int* pixelsOut);
        [DllImport(Lib)] public static extern int rt_pin_pixels(IntPtr ctx, IntPtr pixels, long count);
        [DllImport(Lib)] public static extern int rt_unpin_pixels(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_alloc_pixels(IntPtr ctx, long count, out IntPtr pixels);
        [DllImport(Lib)] public static extern int rt_render_rows_device(IntPtr ctx, ref RtCamera camera, int width, int height, int rowBegin, int rowEnd, int maxDepth, IntPtr devPixelsOut, IntPtr hipStream);
        [DllImport(Lib)] public static extern int rt_render_rows_device_noalloc(IntPtr ctx, ref RtCamera camera, int width, int height, int rowBegin, int rowEnd, int maxDepth, IntPtr devPixelsOut, IntPtr hipStream);
        [DllImport(Lib)] public static extern int rt_render_bands_device(IntPtr ctx, ref RtCamera camera, int width, int height, int bandRows, int part, int parts, int maxDepth, IntPtr devPixelsOut, IntPtr hipStream);
        [DllImport(Lib)] public static extern int rt_band_rows(int height, int bandRows, int part, int parts);
        [DllImport(Lib)] public static extern int rt_set_device_share(IntPtr ctx, int contexts);
        [DllImport(Lib)] public static extern int rt_frame_status(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_fxaa_device(IntPtr ctx, IntPtr devPixelsIn, IntPtr devPixelsOut, int width, int height, IntPtr hipStream);
        [DllImport(Lib)] public static extern int rt_debug_rays(IntPtr ctx, ref RtCamera camera, int width, int height, int row, int xStep, int maxRays, [Out] RtDebugRay[] rays, out int count);
        [DllImport(Lib)] public static extern int rt_get_stats(IntPtr ctx, out RtStats stats);
        [DllImport(Lib)] public static extern int rt_set_profiling(IntPtr ctx, int mode);
        [DllImport(Lib)] public static extern int rt_reload_tuning(IntPtr ctx);

        public static void Check(IntPtr ctx, int status)   // the reference's only error convention: throw new Exception(message), template.cs:102-106
        {
            if (status != 0)
                throw new Exception(Marshal.PtrToStringAnsi(rt_last_error(ctx)));
        }
    }

    // Owns the library context for one Raytracer: the scene flattened the way MyApplication.Init flattens it for
    // the GLSL uniforms (MyApplication.cs:142-176), with lists instead of the arrays of ten, and Surface.pixels
    // pinned for the life of the window (the array is allocated once and never reallocated, surface.cs:14-19).
    unsafe class RtRenderer : IDisposable
    {
        IntPtr ctx;
        GCHandle pixelsHandle;
        int[]? pinned;

        public RtRenderer(int[]? deviceIds = null)
        {
            RtHip.Check(IntPtr.Zero, RtHip.rt_create(out ctx, deviceIds, deviceIds?.Length ?? 0));
        }

        // Call after the host has built (or edited) its scene: MyApplication.Init, template.cs:155.
        public void SetScene(Scene scene)
        {
            int n = scene.primitives.Count, l = scene.lights.Count;
            var types = new int[n]; var checker = new int[n]; var texture = new int[n];
            var positions = new Vector3[n]; var normals = new Vector3[n];
            var diffuse = new Vector4[n]; var glossy = new Vector4[n]; var ambient = new Vector4[n];
            var specular = new float[n]; var radius = new float[n]; var d = new float[n];
            var v0 = new Vector3[n]; var v1 = new Vector3[n]; var v2 = new Vector3[n];
            var lightPositions = new Vector3[l]; var lightColors = new Vector4[l];
            var textures = new List<Texture>();
            for (int i = 0; i < l; i++)
            {
                lightPositions[i] = scene.lights[i].position;
                lightColors[i] = scene.lights[i].color;
            }
            for (int i = 0; i < n; i++)
            {
                Primitive p = scene.primitives[i];
                positions[i] = p.position;
                normals[i] = p.GetNormal(p.position);          // plane / triangle: the constructor's normal (objects.cs:150,233)
                diffuse[i] = p.material.diffuseColor;
                glossy[i] = p.material.glossyColor;
                ambient[i] = p.material.ambientColor;
                specular[i] = p.material.specular;
                checker[i] = p.material.useCheckerboard ? 1 : 0;
                texture[i] = -1;
                if (p is Sphere sphere) { radius[i] = sphere.radius; types[i] = 0; }
                else if (p is Plane plane)
                {
                    d[i] = plane.d; types[i] = 1;
                    if (plane.texture != null)
                    {
                        int slot = textures.IndexOf(plane.texture);
                        if (slot < 0) { slot = textures.Count; textures.Add(plane.texture); }
                        texture[i] = slot;
                    }
                }
                else if (p is Triangle tri) { v0[i] = tri.v0; v1[i] = tri.v1; v2[i] = tri.v2; types[i] = 2; }
            }
            for (int slot = 0; slot < textures.Count; slot++)
                RtHip.Check(ctx, RtHip.rt_set_texture(ctx, slot, textures[slot].ToRgba8(), textures[slot].Width, textures[slot].Height));
            fixed (int* pTypes = types)
            fixed (int* pChecker = checker)
            fixed (int* pTexture = texture)
            fixed (Vector3* pPositions = positions)
            fixed (Vector3* pNormals = normals)
            fixed (Vector4* pDiffuse = diffuse)
            fixed (Vector4* pGlossy = glossy)
            fixed (Vector4* pAmbient = ambient)
            fixed (float* pSpecular = specular)
            fixed (float* pRadius = radius)
            fixed (float* pD = d)
            fixed (Vector3* pV0 = v0)
            fixed (Vector3* pV1 = v1)
            fixed (Vector3* pV2 = v2)
            fixed (Vector3* pLightPositions = lightPositions)
            fixed (Vector4* pLightColors = lightColors)
            {
                var s = new RtScene
                {
                    numPrims = n, numLights = l,
                    primTypes = pTypes,
                    primPositions = (float*)pPositions, primNormals = (float*)pNormals,
                    primDiffuse = (float*)pDiffuse, primGlossy = (float*)pGlossy, primAmbient = (float*)pAmbient,
                    primSpecular = pSpecular, primRadius = pRadius, primD = pD,
                    v0 = (float*)pV0, v1 = (float*)pV1, v2 = (float*)pV2,
                    primCheckerboard = pChecker, primTexture = pTexture,
                    lightPositions = (float*)pLightPositions, lightColors = (float*)pLightColors,
                };
                RtHip.Check(ctx, RtHip.rt_set_scene(ctx, ref s));   // the library copies; no pointer is kept
            }
        }

        // The body of Raytracer.Render() (MyApplication.cs:402-426): blocking, writes every element of screen.pixels.
        public void Render(Camera camera, Surface screen)
        {
            if (!ReferenceEquals(pinned, screen.pixels))
            {
                Unpin();
                pixelsHandle = GCHandle.Alloc(screen.pixels, GCHandleType.Pinned);
                pinned = screen.pixels;
                // Page-locks the array so that the copy to the host is one DMA into it; a system that refuses
                // leaves everything working through the library's staged copy.
                RtHip.rt_pin_pixels(ctx, pixelsHandle.AddrOfPinnedObject(), screen.pixels.Length);
            }
            var cam = new RtCamera
            {
                px = camera.position.X, py = camera.position.Y, pz = camera.position.Z,
                dx = camera.direction.X, dy = camera.direction.Y, dz = camera.direction.Z,
                ux = camera.up.X, uy = camera.up.Y, uz = camera.up.Z,
                fovDeg = camera.fov,
            };
            RtHip.Check(ctx, RtHip.rt_render(ctx, ref cam, screen.width, screen.height, RtHip.ReferenceMaxDepth,
                                             (int*)pixelsHandle.AddrOfPinnedObject()));
        }

        void Unpin()
        {
            if (pinned == null) return;
            RtHip.rt_unpin_pixels(ctx);
            pixelsHandle.Free();
            pinned = null;
        }

        public void Dispose()
        {
            if (ctx == IntPtr.Zero) return;
            Unpin();
            RtHip.rt_destroy(ctx);
            ctx = IntPtr.Zero;
        }
    }
}
