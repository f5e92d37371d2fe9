// librtx_host.so — host side of the MI355X ray-query engine (see include/rt_host.h).
//
// Scene construction, OBJ/BMP loading, the SAH kd-tree build and the flattener.  Written from
// scratch against the *behaviour* of the reference (file:line cited per function); arithmetic that
// decides the shape of the tree is kept op-for-op in binary32 so that, built with
// `-O2 -ffp-contract=off`, the node and leaf-reference streams are identical to the ones the
// reference's KDTree_New produces (checked by tests/test_host_build.py against dumps of the real
// reference, tests/golden/kd.json).
//
// No ray is traced in this file.

#include "rt_host.h"

#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <new>
#include <string>
#include <unordered_map>
#include <thread>
#include <vector>

#define RT_EPSILON 0.0001f // src/common/config.h:15
#define PI32 3.141592653589793238462643383279502884197169399375105820f // src/math/math.h:7
#define INF (__builtin_inff())

static thread_local char g_err[512];

static void set_err(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char* RtHost_LastError(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
// math helpers, in the reference's operation order (src/math/vec.c, src/math/math.c)
// ---------------------------------------------------------------------------------------------
static inline float minf(float x, float y) { return x > y ? y : x; } // math.c:30
static inline float maxf(float x, float y) { return x > y ? x : y; } // math.c:39
static inline vec3  vadd(vec3 a, vec3 b) { return vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline vec3  vsub(vec3 a, vec3 b) { return vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline vec3  vmuls(vec3 v, float s) { return vec3{s * v.x, s * v.y, s * v.z}; } // vec.c:175
static inline float vdot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // vec.c:212
static inline vec3  vcross(vec3 a, vec3 b)                                              // vec.c:217
{
    return vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline vec3 vdivs(vec3 v, float s) { return vmuls(v, 1.0f / s); } // vec.c:198
static inline vec3 vnorm(vec3 v) { return vdivs(v, sqrtf(vdot(v, v))); } // vec.c:241
static inline float elem(const vec3& v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }
static inline float& elem(vec3& v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }

// ---------------------------------------------------------------------------------------------
// images / BMP (src/gfx/image.c) and colour conversion (src/gfx/color.c)
// ---------------------------------------------------------------------------------------------
static float sRGBToLinear(uint8_t channel) // color.c:7-16
{
    float srgb = channel / 255.0f;
    if (srgb < 0.04045f) {
        return srgb / 12.92f;
    } else {
        return powf((srgb + 0.055f) / 1.055f, 2.4f);
    }
}

extern "C" bool ImageRGB_Load_Empty(ImageRGB* img, size_t width, size_t height)
{
    RGB* pix = (RGB*)calloc(width * height ? width * height : 1, sizeof(RGB));
    if (!pix) {
        set_err("ImageRGB_Load_Empty: out of memory");
        return false;
    }
    img->pix        = pix;
    img->res.width  = width;
    img->res.height = height;
    return true;
}

extern "C" void ImageRGB_Unload(ImageRGB* img)
{
    free(img->pix);
    img->pix        = NULL;
    img->res.width  = 0;
    img->res.height = 0;
}

#pragma pack(push, 1)
struct BMPHeader { // image.c:29-53
    uint16_t id;
    uint32_t size;
    uint16_t r0, r1;
    uint32_t pixelArrayOffset;
    uint32_t headerSize;
    int32_t  pixelWidth, pixelHeight;
    uint16_t colorPlanes, bitsPerPixel;
    uint32_t compressionMethod, bitmapSize;
    int32_t  ppmHorizontal, ppmVertical;
    uint32_t numColors, numImportantColors;
};
#pragma pack(pop)
static_assert(sizeof(BMPHeader) == 0x36, "BMP header is 54 bytes");

extern "C" bool ImageRGB_Load_BMP(ImageRGB* img, FILE* fd) // image.c:81-133
{
    BMPHeader h;
    if (fread(&h, sizeof h, 1, fd) != 1) {
        set_err("ImageRGB_Load_BMP: short header");
        return false;
    }
    if (h.id != 0x4D42 || h.bitsPerPixel != 24 || h.compressionMethod != 0 || h.pixelWidth <= 0 || h.pixelHeight <= 0) {
        set_err("ImageRGB_Load_BMP: not a 24-bit uncompressed bottom-up BMP");
        return false;
    }
    size_t   width = (size_t)h.pixelWidth, height = (size_t)h.pixelHeight;
    ImageRGB tmp;
    if (!ImageRGB_Load_Empty(&tmp, width, height)) return false;
    size_t rowBytes = 3 * width;
    size_t pad      = (size_t)(-(ssize_t)rowBytes) % 4;
    if (fseek(fd, h.pixelArrayOffset, SEEK_SET)) {
        ImageRGB_Unload(&tmp);
        set_err("ImageRGB_Load_BMP: seek failed");
        return false;
    }
    // file rows are bottom-up; row 0 of the image becomes the top of the picture (image.c:118)
    for (ssize_t yy = (ssize_t)height - 1; yy >= 0; yy--) {
        if (fread(&tmp.pix[(size_t)yy * width], 3, width, fd) != width || fseek(fd, (long)pad, SEEK_CUR)) {
            ImageRGB_Unload(&tmp);
            set_err("ImageRGB_Load_BMP: short pixel row");
            return false;
        }
    }
    *img = tmp;
    return true;
}

extern "C" bool ImageRGB_Save_BMP(ImageRGB* img, FILE* fd) // image.c:179-237
{
    size_t    rowBytes = 3 * img->res.width;
    size_t    pad      = (size_t)(-(ssize_t)rowBytes) % 4;
    size_t    stride   = rowBytes + pad;
    BMPHeader h;
    memset(&h, 0, sizeof h);
    h.id               = 0x4D42;
    h.size             = (uint32_t)(sizeof h + stride * img->res.height);
    h.pixelArrayOffset = sizeof h;
    h.headerSize       = 40;
    h.pixelWidth       = (int32_t)img->res.width;
    h.pixelHeight      = -(int32_t)img->res.height; // top-down file, rows emitted from y=H-1 (image.c:200,219)
    h.colorPlanes      = 1;
    h.bitsPerPixel     = 24;
    h.bitmapSize       = (uint32_t)(stride * img->res.height);
    h.ppmHorizontal    = (int32_t)img->res.width;
    h.ppmVertical      = (int32_t)img->res.height;
    if (fwrite(&h, sizeof h, 1, fd) != 1) {
        set_err("ImageRGB_Save_BMP: header write failed");
        return false;
    }
    const uint8_t zeros[4] = {0, 0, 0, 0};
    for (ssize_t yy = (ssize_t)img->res.height - 1; yy >= 0; yy--) {
        if (fwrite(&img->pix[(size_t)yy * img->res.width], 3, img->res.width, fd) != img->res.width
            || fwrite(zeros, 1, pad, fd) != pad) {
            set_err("ImageRGB_Save_BMP: row write failed");
            return false;
        }
    }
    return fflush(fd) == 0;
}

extern "C" bool ImageRGB_Load_BMP_Path(ImageRGB* img, const char* path)
{
    FILE* fd = fopen(path, "rb");
    if (!fd) {
        set_err("cannot open %s", path);
        return false;
    }
    bool ok = ImageRGB_Load_BMP(img, fd);
    fclose(fd);
    return ok;
}

extern "C" bool ImageRGB_Save_BMP_Path(ImageRGB* img, const char* path)
{
    FILE* fd = fopen(path, "wb+");
    if (!fd) {
        set_err("cannot open %s for writing", path);
        return false;
    }
    bool ok = ImageRGB_Save_BMP(img, fd);
    fclose(fd);
    return ok;
}

// ---------------------------------------------------------------------------------------------
// textures (src/gfx/texture.c) and skybox (src/world/skybox.c:12-64)
// ---------------------------------------------------------------------------------------------
struct Texture {
    ImageColor* image;
};

extern "C" Texture* Texture_New(void) { return (Texture*)calloc(1, sizeof(Texture)); }

static bool texture_alloc(Texture* tex, size_t w, size_t h)
{
    ImageColor* img = (ImageColor*)calloc(1, sizeof(ImageColor));
    if (!img) return false;
    img->pix = (Color*)calloc(w * h ? w * h : 1, sizeof(Color));
    if (!img->pix) {
        free(img);
        return false;
    }
    img->res.width  = w;
    img->res.height = h;
    if (tex->image) {
        free(tex->image->pix);
        free(tex->image);
    }
    tex->image = img;
    return true;
}

extern "C" bool Texture_Import_BMP(Texture* tex, FILE* fd) // texture.c:15-47
{
    ImageRGB bmp;
    if (!ImageRGB_Load_BMP(&bmp, fd)) return false;
    if (!texture_alloc(tex, bmp.res.width, bmp.res.height)) {
        ImageRGB_Unload(&bmp);
        set_err("Texture_Import_BMP: out of memory");
        return false;
    }
    size_t n = bmp.res.width * bmp.res.height;
    for (size_t i = 0; i < n; i++) { // Color_FromRGB, color.c:73-80
        tex->image->pix[i] = Color{sRGBToLinear(bmp.pix[i].r), sRGBToLinear(bmp.pix[i].g), sRGBToLinear(bmp.pix[i].b)};
    }
    ImageRGB_Unload(&bmp);
    return true;
}

extern "C" bool Texture_Import_BMP_Path(Texture* tex, const char* path)
{
    FILE* fd = fopen(path, "rb");
    if (!fd) {
        set_err("cannot open %s", path);
        return false;
    }
    bool ok = Texture_Import_BMP(tex, fd);
    fclose(fd);
    return ok;
}

extern "C" bool Texture_Import_Color(Texture* tex, Color color) // texture.c:49-70
{
    if (!texture_alloc(tex, 1, 1)) {
        set_err("Texture_Import_Color: out of memory");
        return false;
    }
    tex->image->pix[0] = color;
    return true;
}

extern "C" bool Texture_Import_Pixels(Texture* tex, size_t width, size_t height, const float* rgb)
{
    if (!width || !height || !rgb) {
        set_err("Texture_Import_Pixels: empty image");
        return false;
    }
    if (!texture_alloc(tex, width, height)) {
        set_err("Texture_Import_Pixels: out of memory");
        return false;
    }
    memcpy(tex->image->pix, rgb, width * height * sizeof(Color));
    return true;
}

extern "C" void Texture_Delete(Texture* tex)
{
    if (!tex) return;
    if (tex->image) {
        free(tex->image->pix);
        free(tex->image);
    }
    free(tex);
}

extern "C" const ImageColor* Texture_Image(const Texture* tex) { return tex ? tex->image : NULL; }

extern "C" void Skybox_Delete(Skybox* skybox)
{
    if (!skybox) return;
    for (int i = 0; i < 6; i++) {
        if (skybox->tex[i]) Texture_Delete(skybox->tex[i]);
    }
    free(skybox);
}

extern "C" Skybox* Skybox_Import_BMP(const char* folder_path) // skybox.c:21-64
{
    static const char* names[6] = {"xpos.bmp", "xneg.bmp", "ypos.bmp", "yneg.bmp", "zpos.bmp", "zneg.bmp"};
    Skybox*            sky      = (Skybox*)calloc(1, sizeof(Skybox));
    if (!sky) return NULL;
    for (int i = 0; i < 6; i++) {
        std::string path = std::string(folder_path) + "/" + names[i];
        Texture*    tex  = Texture_New();
        if (!tex || !Texture_Import_BMP_Path(tex, path.c_str())) {
            Texture_Delete(tex);
            Skybox_Delete(sky);
            return NULL;
        }
        sky->tex[i] = tex;
    }
    return sky;
}

extern "C" Skybox* Skybox_New_Color(Color color)
{
    Skybox* sky = (Skybox*)calloc(1, sizeof(Skybox));
    if (!sky) return NULL;
    for (int i = 0; i < 6; i++) {
        sky->tex[i] = Texture_New();
        if (!sky->tex[i] || !Texture_Import_Color(sky->tex[i], color)) {
            Skybox_Delete(sky);
            return NULL;
        }
    }
    return sky;
}

// ---------------------------------------------------------------------------------------------
// materials (constructors only; the Bounce functions live on the GPU)
// p[] slots: 0 subsurface 1 specular 2 roughness 3 specular_tint 4 anistropic 5 sheen_tint
//            6 clearcoat_gloss 7 eta 8 w.clearcoat 9 w.sheen 10 w.metallic 11 w.specular
// ---------------------------------------------------------------------------------------------
static Material mat_zero(uint32_t type, Texture* tex)
{
    Material m;
    memset(&m, 0, sizeof m);
    m.type   = type;
    m.albedo = tex;
    return m;
}

extern "C" Material Material_Diffuse_Make(Texture* tex) { return mat_zero(RTX_MAT_DIFFUSE, tex); }

extern "C" Material Material_Metal_Make(Texture* tex, float fuzz)
{
    Material m = mat_zero(RTX_MAT_METAL, tex);
    m.p[0]     = fuzz;
    return m;
}

extern "C" Material Material_Dielectric_Make(Texture* tex, float refractiveIndex)
{
    Material m = mat_zero(RTX_MAT_DIELECTRIC, tex);
    m.p[0]     = refractiveIndex;
    return m;
}

extern "C" Material Material_DiffuseLight_Make(Texture* tex, float brightness)
{
    Material m = mat_zero(RTX_MAT_DIFFUSE_LIGHT, tex);
    m.p[0]     = brightness;
    return m;
}

extern "C" Material Material_Skybox_Make(Skybox* skybox)
{
    Material m = mat_zero(RTX_MAT_SKYBOX, NULL);
    m.skybox   = skybox;
    return m;
}

extern "C" Material Material_Disney_Diffuse_Make(Texture* albedo, float roughness, float subsurface)
{
    Material m = mat_zero(RTX_MAT_DISNEY_DIFFUSE, albedo);
    m.p[2]     = roughness;
    m.p[0]     = subsurface;
    return m;
}

extern "C" Material Material_Disney_Metal_Make(Texture* albedo, float roughness, float anistropic)
{
    Material m = mat_zero(RTX_MAT_DISNEY_METAL, albedo);
    m.p[2]     = roughness;
    m.p[4]     = anistropic;
    m.p[10]    = 1.0f; // weights.metallic = 1 (materials.c:334)
    return m;
}

extern "C" Material Material_Disney_Clearcoat_Make(float gloss)
{
    Material m = mat_zero(RTX_MAT_DISNEY_CLEARCOAT, NULL);
    m.p[6]     = gloss;
    return m;
}

extern "C" Material Material_Disney_Glass_Make(Texture* albedo, float roughness, float anistropic, float eta)
{
    Material m = mat_zero(RTX_MAT_DISNEY_GLASS, albedo);
    m.p[2]     = roughness;
    m.p[4]     = anistropic;
    m.p[7]     = eta;
    return m;
}

extern "C" Material Material_Disney_Sheen_Make(Texture* albedo, float sheen_tint)
{
    Material m = mat_zero(RTX_MAT_DISNEY_SHEEN, albedo);
    m.p[5]     = sheen_tint;
    return m;
}

extern "C" Material Material_Disney_BSDF_Make(Texture* albedo, float subsurface, float specular, float roughness,
                                              float specular_tint, float anistropic, float sheen_tint,
                                              float clearcoat_gloss, float eta, float weight_sheen,
                                              float weight_clearcoat, float weight_specular, float weight_metallic)
{
    Material m = mat_zero(RTX_MAT_DISNEY_BSDF, albedo);
    m.p[0]     = subsurface;
    m.p[1]     = specular;
    m.p[2]     = roughness;
    m.p[3]     = specular_tint;
    m.p[4]     = anistropic;
    m.p[5]     = sheen_tint;
    m.p[6]     = clearcoat_gloss;
    m.p[7]     = eta;
    m.p[8]     = weight_clearcoat;
    m.p[9]     = weight_sheen;
    m.p[10]    = weight_metallic;
    m.p[11]    = weight_specular;
    return m;
}

// ---------------------------------------------------------------------------------------------
// surfaces: constructors + bounding boxes (src/rt/surfaces.c); HitAt lives on the GPU
// ---------------------------------------------------------------------------------------------
extern "C" Surface Surface_Sphere_Make(point3 center, float radius)
{
    Surface s;
    memset(&s, 0, sizeof s);
    s.type     = SURFACE_SPHERE;
    s.sphere.c = center;
    s.sphere.r = radius;
    return s;
}

extern "C" Surface Surface_Triangle_Make(Vertex v0, Vertex v1, Vertex v2)
{
    Surface s;
    memset(&s, 0, sizeof s);
    s.type            = SURFACE_TRIANGLE;
    s.triangle.vtx[0] = v0;
    s.triangle.vtx[1] = v1;
    s.triangle.vtx[2] = v2;
    return s;
}

extern "C" Surface Surface_Plane_Make(point3 point, vec3 normal)
{
    Surface s;
    memset(&s, 0, sizeof s);
    s.type         = SURFACE_PLANE;
    s.plane.normal = normal;
    s.plane.point  = point;
    return s;
}

extern "C" Triangle Triangle_MakeSimple(point3 v0, point3 v1, point3 v2) // surfaces.c:91-102
{
    vec3     n = vcross(vsub(v1, v0), vsub(v2, v0));
    Triangle t;
    t.vtx[0] = Vertex{v0, n, vec2{0.0f, 0.0f}};
    t.vtx[1] = Vertex{v1, n, vec2{0.0f, 0.0f}};
    t.vtx[2] = Vertex{v2, n, vec2{0.0f, 0.0f}};
    return t;
}

static BoundingBox Sphere_BoundingBox(const Sphere* s) // surfaces.c:23-37
{
    float r = s->r + RT_EPSILON;
    vec3  v = vec3{r, r, r};
    return BoundingBox{vsub(s->c, v), vadd(s->c, v)};
}

static BoundingBox Triangle_BoundingBox(const Triangle* t) // surfaces.c:114-128
{
    BoundingBox box;
    for (int a = 0; a < 3; a++) {
        float p0 = elem(t->vtx[0].pos, a), p1 = elem(t->vtx[1].pos, a), p2 = elem(t->vtx[2].pos, a);
        elem(box.min, a) = minf(minf(p0, p1), p2) - RT_EPSILON;
        elem(box.max, a) = maxf(maxf(p0, p1), p2) + RT_EPSILON;
    }
    return box;
}

static BoundingBox Plane_BoundingBox(const Plane* p) // surfaces.c:192-216
{
    for (int a = 0; a < 3; a++) {
        int next = (a + 1) % 3, prev = (a + 2) % 3;
        if (elem(p->normal, next) == 0.0f && elem(p->normal, prev) == 0.0f) {
            BoundingBox box;
            elem(box.min, prev) = -INF;
            elem(box.min, a)    = -RT_EPSILON;
            elem(box.min, next) = -INF;
            elem(box.max, prev) = INF;
            elem(box.max, a)    = RT_EPSILON;
            elem(box.max, next) = INF;
            return box;
        }
    }
    return BoundingBox{vec3{-INF, -INF, -INF}, vec3{INF, INF, INF}};
}

extern "C" BoundingBox Surface_BoundingBox(Surface* s) // object.c:26-43
{
    switch (s->type) {
        case SURFACE_SPHERE: return Sphere_BoundingBox(&s->sphere);
        case SURFACE_TRIANGLE: return Triangle_BoundingBox(&s->triangle);
        default: return Plane_BoundingBox(&s->plane);
    }
}

extern "C" bool Surface_Bounded(Surface* s) { return s->type != SURFACE_PLANE; } // object.c:45-62

// ---------------------------------------------------------------------------------------------
// SAH kd-tree build (src/rt/accelerators/kdtree.c:137-444)
//
// Output is produced directly in the 8-byte node form of rtx_scene_desc; `refs` holds, per leaf
// entry, the index of the object in the array handed to KDTree_New (the reference stores Object*).
// The visiting order (parent slot, RIGHT subtree, then LEFT subtree) and every f32 operation of
// the SAH search are the reference's.  The only algorithmic change is how the per-candidate
// primitive counts are obtained: instead of one pass over the primitives per candidate plane
// (kdtree.c:254-265, 93 passes per node) one pass bins every box against the sorted candidate
// list; the integer counts, and therefore the f32 costs, are identical.
// ---------------------------------------------------------------------------------------------
#define MIN_LEAF_LOAD 4
#define NUM_BUCKETS 32
#define INTERSECT_COST (1.0f)
#define RIGHT_NODE_RELATIVE_COST (0.95f)
#define EMPTY_BONUS (0.5f)
#define TRAVERSAL_COST (1.0f)
#define LEFT_NODE_RELATIVE_COST (1.0f + (1.0f - RIGHT_NODE_RELATIVE_COST))
#define KD_LEAF 3u
#define KD_MAX_INDEX ((1u << 30) - 1) // kdtree.c:412-413

struct KDTree {
    std::vector<uint32_t> w0, w1; // node words
    std::vector<uint32_t> refs;   // object indices (kd order)
    BoundingBox           worldBox;
    size_t                rootIndex;
    size_t                maxDepth;
    size_t                depthReached;
    bool                  failed;
};

struct KDBuild {
    KDTree*                   tree;
    const std::vector<BoundingBox>* boxes;
};

static float SurfaceArea(const BoundingBox& b) // kdtree.c:198-205
{
    float x = b.max.x - b.min.x, y = b.max.y - b.min.y, z = b.max.z - b.min.z;
    return 2 * x * y + 2 * x * z + 2 * y * z;
}

static uint32_t float_bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

static size_t BuildLeafNode(KDBuild& b, const std::vector<uint32_t>& vect) // kdtree.c:173-196
{
    KDTree* t   = b.tree;
    size_t  idx = t->w0.size();
    t->w0.push_back((uint32_t)vect.size());
    t->w1.push_back((KD_LEAF << 30) | (uint32_t)t->refs.size());
    t->refs.insert(t->refs.end(), vect.begin(), vect.end());
    return idx;
}

static size_t BuildNode(KDBuild& b, const std::vector<uint32_t>& vect, BoundingBox container, size_t depth, size_t level)
{
    KDTree* t = b.tree;
    if (t->failed) return 0;
    if (t->w0.size() > KD_MAX_INDEX || t->refs.size() > KD_MAX_INDEX) { // kdtree.c:280 (2^30 cap of :412)
        t->failed = true;
        set_err("KDTree_New: more than 2^30 nodes or leaf references");
        return 0;
    }
    if (level > t->depthReached) t->depthReached = level;

    const size_t n = vect.size();
    if (n <= MIN_LEAF_LOAD || depth == 0) { // kdtree.c:286
        return BuildLeafNode(b, vect);
    }

    const std::vector<BoundingBox>& boxes = *b.boxes;

    float bestSplit = 0.0f;
    int   bestAxis  = 0;
    float bestSAH   = INF;

    const float parentSA = SurfaceArea(container);

    for (int axis = 0; axis < 3; axis++) { // kdtree.c:295-309
        const float cmin = elem(container.min, axis), cmax = elem(container.max, axis);
        const float stride = (cmax - cmin) / NUM_BUCKETS;

        // candidate planes, generated with the reference's f32 accumulation loop
        float cand[2 * NUM_BUCKETS + 2];
        int   k = 0;
        if (!(stride > 0.0f) || cmin + stride == cmin) {
            // the reference would spin forever here (bucket += 0); no such node arises from
            // padded boxes in practice, so just offer no candidate on this axis
            continue;
        }
        for (float bucket = cmin + stride; bucket <= cmax - stride; bucket += stride) {
            if (k == 2 * NUM_BUCKETS + 2) break;
            cand[k++] = bucket;
        }
        if (k == 0) continue;

        // cntMaxLt[j] = #{boxes: max < cand[j]},  cntMinGt[j] = #{boxes: min > cand[j]}
        // via histograms of a = #{j: cand[j] <= max} and g = #{j: cand[j] < min}
        uint32_t histA[2 * NUM_BUCKETS + 4] = {0}, histG[2 * NUM_BUCKETS + 4] = {0};
        const float invStride = 1.0f / stride;
        for (size_t ii = 0; ii < n; ii++) {
            const BoundingBox& bx = boxes[vect[ii]];
            float bmax = elem(bx.max, axis), bmin = elem(bx.min, axis);
            // a: count of candidates <= bmax (guess, then fix with exact compares)
            int a = (int)((bmax - cand[0]) * invStride) + 1;
            if (!(a >= 0)) a = 0;
            if (a > k) a = k;
            while (a < k && cand[a] <= bmax) a++;
            while (a > 0 && !(cand[a - 1] <= bmax)) a--;
            histA[a]++;
            int g = (int)((bmin - cand[0]) * invStride) + 1;
            if (!(g >= 0)) g = 0;
            if (g > k) g = k;
            while (g < k && cand[g] < bmin) g++;
            while (g > 0 && !(cand[g - 1] < bmin)) g--;
            histG[g]++;
        }
        // box.max < cand[j]  <=>  a <= j ;  box.min > cand[j]  <=>  g >= j+1
        size_t maxLt = 0;                // running #{a <= j}
        size_t minGt = n;                // running #{g >= j+1}
        for (int j = 0; j < k; j++) {
            maxLt += histA[j];
            minGt -= histG[j];
            // kdtree.c:254-265: left-only if max<split; else right-only if min>split; else both.
            // padded boxes have min < max so the two tests are exclusive.
            size_t leftPrims  = n - minGt;
            size_t rightPrims = n - maxLt;

            const float split = cand[j];
            BoundingBox lb = container, rb = container; // SplitBox, kdtree.c:232-242
            elem(lb.max, axis) = split;
            elem(rb.min, axis) = split;
            // ComputeSplitSAH, kdtree.c:267-271
            float leftCost   = (SurfaceArea(lb) / parentSA) * leftPrims * INTERSECT_COST * LEFT_NODE_RELATIVE_COST;
            float rightCost  = (SurfaceArea(rb) / parentSA) * rightPrims * INTERSECT_COST * RIGHT_NODE_RELATIVE_COST;
            float emptyBonus = (leftPrims == 0 || rightPrims == 0) ? EMPTY_BONUS : 0;
            float SAH        = TRAVERSAL_COST + (1.0f - emptyBonus) * (leftCost + rightCost);
            if (SAH < bestSAH) {
                bestSAH   = SAH;
                bestAxis  = axis;
                bestSplit = split;
            }
        }
    }

    float parentSAH = n * INTERSECT_COST; // kdtree.c:313
    if (parentSAH <= bestSAH) {
        return BuildLeafNode(b, vect);
    }

    std::vector<uint32_t> leftVect, rightVect; // kdtree.c:331-347
    leftVect.reserve(n);
    rightVect.reserve(n);
    for (size_t ii = 0; ii < n; ii++) {
        const BoundingBox& bx = boxes[vect[ii]];
        if (elem(bx.max, bestAxis) < bestSplit) {
            leftVect.push_back(vect[ii]);
        } else if (elem(bx.min, bestAxis) > bestSplit) {
            rightVect.push_back(vect[ii]);
        } else {
            leftVect.push_back(vect[ii]);
            rightVect.push_back(vect[ii]);
        }
    }

    size_t      new_depth = depth == 0 ? 0 : depth - 1;
    BoundingBox lb = container, rb = container;
    elem(lb.max, bestAxis) = bestSplit;
    elem(rb.min, bestAxis) = bestSplit;

    // BuildParentNode, kdtree.c:137-171: reserve the parent slot, RIGHT subtree first so that the
    // right child is parent+1, then the LEFT subtree
    size_t parentIndex = t->w0.size();
    t->w0.push_back(0);
    t->w1.push_back(0);
    BuildNode(b, rightVect, rb, new_depth, level + 1);
    std::vector<uint32_t>().swap(rightVect);
    size_t leftNode = BuildNode(b, leftVect, lb, new_depth, level + 1);
    t->w0[parentIndex] = float_bits(bestSplit);
    t->w1[parentIndex] = ((uint32_t)bestAxis << 30) | (uint32_t)leftNode;
    return parentIndex;
}

// what KDTree_New prepares before the recursion: depth bound, primitive boxes, world box (kdtree.c:405-435)
// `boxes` = CreateKDBBs (kdtree.c:385-403), filled by the caller: from Objects for the reference-named entry
// points, straight from the scene's triangle pool for Scene_Prepare
static KDTree* KDTree_Inputs(const std::vector<BoundingBox>& boxes)
{
    const size_t len   = boxes.size();
    KDTree* tree       = new KDTree();
    tree->failed       = false;
    tree->depthReached = 0;
    tree->maxDepth     = (size_t)(8.0 + 1.8 * log2((double)len));

    BoundingBox world = boxes[0]; // BoxBoundingAll, kdtree.c:207-230
    for (size_t i = 1; i < len; i++) {
        for (int a = 0; a < 3; a++) {
            if (elem(boxes[i].min, a) < elem(world.min, a)) elem(world.min, a) = elem(boxes[i].min, a);
            if (elem(boxes[i].max, a) > elem(world.max, a)) elem(world.max, a) = elem(boxes[i].max, a);
        }
    }
    tree->worldBox = world;
    return tree;
}

static void Object_Boxes(Object* objs, size_t len, std::vector<BoundingBox>& boxes)
{
    boxes.resize(len);
    for (size_t i = 0; i < len; i++) boxes[i] = Surface_BoundingBox(&objs[i].surface);
}

static KDTree* kd_from_boxes_external(const std::vector<BoundingBox>& boxes, KDExternalBuild build, KDExternalFree release, int device, double* ms_device)
{
    const size_t       len  = boxes.size();
    KDTree*            tree = KDTree_Inputs(boxes);
    std::vector<float> flat(6 * len);
    for (size_t i = 0; i < len; i++) {
        for (int a = 0; a < 3; a++) flat[6 * i + a] = elem(boxes[i].min, a), flat[6 * i + 3 + a] = elem(boxes[i].max, a);
    }
    float world[6];
    for (int a = 0; a < 3; a++) world[a] = elem(tree->worldBox.min, a), world[3 + a] = elem(tree->worldBox.max, a);
    rtx_kd_tree out;
    memset(&out, 0, sizeof out);
    const int rc = build(device, flat.data(), len, tree->maxDepth, world, &out);
    if (rc != 0) { // no silent fall-back to the host recursion: the caller asked for the external builder
        set_err("KDTree_New_External: the external builder failed with code %d", rc);
        delete tree;
        return NULL;
    }
    tree->w0.assign(out.node_w0, out.node_w0 + out.n_nodes);
    tree->w1.assign(out.node_w1, out.node_w1 + out.n_nodes);
    tree->refs.assign(out.leaf_ref, out.leaf_ref + out.n_refs);
    tree->rootIndex    = out.root_index;
    tree->depthReached = out.depth_reached;
    if (ms_device) *ms_device = out.ms_device;
    release(&out);
    return tree;
}

extern "C" KDTree* KDTree_New_External(Object* objs, size_t len, KDExternalBuild build, KDExternalFree release, int device, double* ms_device)
{
    if (!objs || len == 0 || !build || !release) {
        set_err("KDTree_New_External: no objects or no builder");
        return NULL;
    }
    std::vector<BoundingBox> boxes;
    Object_Boxes(objs, len, boxes);
    return kd_from_boxes_external(boxes, build, release, device, ms_device);
}

static KDTree* kd_from_boxes_host(std::vector<BoundingBox>& boxes) // kdtree.c:405-444 after CreateKDBBs
{
    const size_t      len   = boxes.size();
    KDTree*           tree  = KDTree_Inputs(boxes);
    const BoundingBox world = tree->worldBox;

    std::vector<uint32_t> all(len);
    for (size_t i = 0; i < len; i++) all[i] = (uint32_t)i;

    KDBuild b{tree, &boxes};
    tree->rootIndex = BuildNode(b, all, world, tree->maxDepth, 0);
    if (tree->failed) {
        delete tree;
        return NULL;
    }
    return tree;
}

extern "C" KDTree* KDTree_New(Object* objs, size_t len) // kdtree.c:405-444
{
    if (!objs || len == 0) {
        set_err("KDTree_New: no objects");
        return NULL;
    }
    std::vector<BoundingBox> boxes;
    Object_Boxes(objs, len, boxes);
    return kd_from_boxes_host(boxes);
}

extern "C" void KDTree_Delete(KDTree* tree) { delete tree; }

static uint64_t fnv1a(const void* data, size_t bytes, uint64_t h = 1469598103934665603ull)
{
    const uint8_t* p = (const uint8_t*)data;
    for (size_t i = 0; i < bytes; i++) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h;
}

extern "C" void KDTree_Stats(const KDTree* tree, KDTreeStats* out)
{
    memset(out, 0, sizeof *out);
    out->nodes           = tree->w0.size();
    out->refs            = tree->refs.size();
    out->max_depth_bound = tree->maxDepth;
    out->depth_reached   = tree->depthReached;
    uint64_t h           = 1469598103934665603ull;
    for (size_t i = 0; i < tree->w0.size(); i++) {
        if ((tree->w1[i] >> 30) == KD_LEAF) {
            out->leaves++;
            if (tree->w0[i] == 0) out->empty_leaves++;
            if (tree->w0[i] > out->max_leaf_len) out->max_leaf_len = tree->w0[i];
        }
        uint32_t pair[2] = {tree->w0[i], tree->w1[i]};
        h = fnv1a(pair, 8, h);
    }
    out->fnv1a_nodes = h;
    out->fnv1a_refs  = fnv1a(tree->refs.data(), tree->refs.size() * 4);
}

// ---------------------------------------------------------------------------------------------
// scene (src/world/scene.c)
// ---------------------------------------------------------------------------------------------
// The reference keeps three vectors of 112-byte Objects (all, bounded, unbounded: scene.c:16-22, 145-166), i.e.
// 224 bytes and two copies per triangle before anything is flattened.  Here a triangle goes from the importer's
// Triangle straight into one compact pool (96-byte Triangle + its material), spheres and planes — a handful —
// stay Objects, and `order` remembers the order of the Scene_Add_Object calls; Scene_Prepare derives the
// reference's kdObjects / unboundObjs sequences from it as index lists.  The kdObjects index
// of a primitive — the prim_id reported across the C ABI — is its position in `kdSeq`, as in the reference.
#define SCENE_OTHER 0x80000000u

// body(lo, hi) over [0, n) in contiguous ranges on the host's cores (RTX_HOST_THREADS, default up to 16); small n
// runs on the caller.  Used where a scene build touches every triangle once: results do not depend on the split.
static unsigned host_threads()
{
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    if (n > 16) n = 16;
    if (const char* env = getenv("RTX_HOST_THREADS")) {
        const int v = atoi(env);
        if (v >= 1 && v <= 64) n = (unsigned)v;
    }
    return n;
}
template <class F> static void parallel_ranges(size_t n, F&& body)
{
    const unsigned T = n < (1u << 16) ? 1 : host_threads();
    if (T == 1) {
        body((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned k = 1; k < T; k++) th.emplace_back([&, k] { body(n * k / T, n * (k + 1) / T); });
    body((size_t)0, n / T);
    for (auto& t : th) t.join();
}
// vector whose resize() leaves new elements uninitialised: the triangle pools are resized and then filled by
// several threads, and a value-initialising resize would first zero 100 MB on one
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
typedef std::vector<Triangle, NoInitAlloc<Triangle>> TriangleVec;

struct Scene {
    Skybox*                skybox;
    TriangleVec            tris;
    std::vector<Material*> triMat;
    std::vector<Object>    others;           // spheres and planes
    std::vector<uint32_t>  order;            // per added object: index into tris, or SCENE_OTHER | index into others
    std::vector<uint32_t>  kdSeq, unboundSeq; // after Scene_Prepare: entries of `order` / indices into others
    KDTree*                kdTree;
};

extern "C" Scene* Scene_New(Skybox* skybox)
{
    Scene* s  = new Scene();
    s->skybox = skybox;
    s->kdTree = NULL;
    return s;
}

extern "C" void Scene_Delete(Scene* scene)
{
    if (!scene) return;
    if (scene->kdTree) KDTree_Delete(scene->kdTree);
    delete scene;
}

extern "C" bool Scene_Add_Object(Scene* scene, Object* obj)
{
    if (scene->order.size() >= 0x7FFFFFFFu) {
        set_err("Scene_Add_Object: more than 2^31 objects");
        return false;
    }
    if (obj->surface.type == SURFACE_TRIANGLE) {
        scene->order.push_back((uint32_t)scene->tris.size());
        scene->tris.push_back(obj->surface.triangle);
        scene->triMat.push_back(obj->material);
    } else {
        scene->order.push_back(SCENE_OTHER | (uint32_t)scene->others.size());
        scene->others.push_back(*obj);
    }
    return true;
}

// scene.c:145-166 up to the tree: the bounded / unbounded split and CreateKDBBs
static void Scene_Split(Scene* scene, std::vector<BoundingBox>& boxes)
{
    scene->kdSeq.clear();
    scene->unboundSeq.clear();
    if (scene->kdTree) KDTree_Delete(scene->kdTree);
    scene->kdTree = NULL;
    scene->kdSeq.reserve(scene->order.size());
    for (uint32_t e : scene->order) {
        if (!(e & SCENE_OTHER) || Surface_Bounded(&scene->others[e & ~SCENE_OTHER].surface)) {
            scene->kdSeq.push_back(e);
        } else {
            scene->unboundSeq.push_back(e & ~SCENE_OTHER);
        }
    }
    boxes.resize(scene->kdSeq.size());
    parallel_ranges(boxes.size(), [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            const uint32_t e = scene->kdSeq[i];
            boxes[i]         = (e & SCENE_OTHER) ? Surface_BoundingBox(&scene->others[e & ~SCENE_OTHER].surface) : Triangle_BoundingBox(&scene->tris[e]);
        }
    });
}

extern "C" bool Scene_Prepare(Scene* scene) // scene.c:145-166
{
    std::vector<BoundingBox> boxes;
    Scene_Split(scene, boxes);
    if (!boxes.empty()) {
        scene->kdTree = kd_from_boxes_host(boxes);
        if (!scene->kdTree) return false;
    }
    return true;
}

extern "C" bool Scene_Prepare_External(Scene* scene, KDExternalBuild build, KDExternalFree release, int device, double* ms_device)
{
    if (!build || !release) {
        set_err("Scene_Prepare_External: no builder");
        return false;
    }
    std::vector<BoundingBox> boxes;
    Scene_Split(scene, boxes);
    if (ms_device) *ms_device = 0.0;
    if (!boxes.empty()) {
        scene->kdTree = kd_from_boxes_external(boxes, build, release, device, ms_device);
        if (!scene->kdTree) return false;
    }
    return true;
}

extern "C" size_t        Scene_Object_Count(const Scene* scene) { return scene->order.size(); }
extern "C" const KDTree* Scene_KDTree(const Scene* scene) { return scene->kdTree; }

// ---------------------------------------------------------------------------------------------
// mesh / OBJ import (src/gfx/mesh.c)
// ---------------------------------------------------------------------------------------------
struct Mesh {
    point3                origin;
    float                 scale;
    Material*             material;
    TriangleVec           polys;
};

extern "C" Mesh* Mesh_New(void)
{
    Mesh* m     = new Mesh();
    m->origin   = point3{0, 0, 0};
    m->scale    = 0.0f; // calloc'd in the reference (mesh.c:27): scale must be set by the caller
    m->material = NULL;
    return m;
}

extern "C" void   Mesh_Delete(Mesh* mesh) { delete mesh; }
extern "C" void   Mesh_Set_Material(Mesh* mesh, Material* material) { mesh->material = material; }
extern "C" void   Mesh_Set_Origin(Mesh* mesh, point3 origin) { mesh->origin = origin; }
extern "C" void   Mesh_Set_Scale(Mesh* mesh, float scale) { mesh->scale = scale; }
extern "C" size_t Mesh_Triangle_Count(const Mesh* mesh) { return mesh->polys.size(); }
extern "C" const Triangle* Mesh_Triangles(const Mesh* mesh) { return mesh->polys.data(); }

extern "C" void Mesh_AddToScene(Mesh* mesh, Scene* scene) // mesh.c:67-84
{
    // same transform, pos*scale + origin (mesh.c:74-76), written straight into the scene's triangle pool: no
    // 112-byte Object is built per triangle
    const size_t n    = mesh->polys.size();
    const size_t base = scene->tris.size();
    if (scene->order.size() + n >= 0x7FFFFFFFu) {
        set_err("Mesh_AddToScene: more than 2^31 objects");
        return;
    }
    scene->tris.resize(base + n);
    scene->triMat.resize(base + n, mesh->material);
    scene->order.resize(scene->order.size() + n);
    uint32_t* ord = scene->order.data() + scene->order.size() - n;
    parallel_ranges(n, [&](size_t lo, size_t hi) {
        for (size_t ii = lo; ii < hi; ii++) {
            Triangle tri = mesh->polys[ii];
            for (int jj = 0; jj < 3; jj++) {
                tri.vtx[jj].pos = vadd(vmuls(tri.vtx[jj].pos, mesh->scale), mesh->origin);
            }
            scene->tris[base + ii] = tri;
            ord[ii]                = (uint32_t)(base + ii);
        }
    });
}

// Fast paths of the importer.  The reference parses every line with up to three sscanf calls and every face
// corner with up to four (mesh.c:104-153); on a 1.1 M-triangle file that is most of the scene build once the
// kd-tree is built on the device.  A line or token of the plain shape every exporter writes ("v x y z" with
// plain decimal numbers, "a/b/c" with plain digits) is parsed directly — strtof is what sscanf's %f calls, so
// the floats are the same bits —; anything else (signs, hex floats, inf/nan, trailing text, odd spacing) takes
// the sscanf path unchanged, so the quirks stay the reference's.
static bool plain_number_at(const char* p)
{
    if (*p == '-') p++;
    return (*p >= '0' && *p <= '9') || (*p == '.' && p[1] >= '0' && p[1] <= '9');
}
static bool is_blank(char c) { return c == ' ' || c == '\t'; }
static bool fast_floats(const char* p, int n, float* out) // p just past the keyword; n plain numbers, blank-separated
{
    for (int i = 0; i < n; i++) {
        if (!is_blank(*p)) return false;
        while (is_blank(*p)) p++;
        if (!plain_number_at(p)) return false;
        const char* q = p;
        if (*q == '-') q++;
        while ((*q >= '0' && *q <= '9') || *q == '.') q++;
        if (*q == 'e' || *q == 'E') {
            const char* r = q + 1;
            if (*r == '-' || *r == '+') r++;
            if (!(*r >= '0' && *r <= '9')) return false;
            while (*r >= '0' && *r <= '9') r++;
            q = r;
        }
        // the number must end where the next blank / the line end begins: "1.0abc" or "0x10" go to sscanf
        if (!(is_blank(*q) || *q == '\n' || *q == '\r' || *q == 0)) return false;
        // Clinger's fast path: up to 15 significant digits and |exponent| <= 22 make mantissa and power of ten
        // exact doubles, their product or quotient a correctly rounded double; rounding that to float is the
        // correctly rounded float — what strtof returns — unless the double sits within an ulp of the midpoint
        // of two floats (or outside the normal float range): those, and longer numbers, go to strtof itself.
        {
            const char* c = p;
            bool        neg = false;
            if (*c == '-') neg = true, c++;
            uint64_t m = 0;
            int      nd = 0, e10 = 0;
            bool     ok = true, dot = false, lead = true;
            for (; (*c >= '0' && *c <= '9') || *c == '.'; c++) {
                if (*c == '.') {
                    if (dot) { ok = false; break; }
                    dot = true;
                    continue;
                }
                if (lead && *c == '0') { if (dot) e10--; continue; }
                lead = false;
                if (++nd > 15) { ok = false; break; }
                m = m * 10 + (uint64_t)(*c - '0');
                if (dot) e10--;
            }
            if (ok && (*c == 'e' || *c == 'E')) {
                c++;
                bool eneg = false;
                if (*c == '-') eneg = true, c++;
                else if (*c == '+') c++;
                int ev = 0, ed = 0;
                for (; *c >= '0' && *c <= '9'; c++) { if (++ed > 4) { ok = false; break; } ev = ev * 10 + (*c - '0'); }
                e10 += eneg ? -ev : ev;
            }
            static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
            if (ok && c == q && e10 >= -22 && e10 <= 22) {
                const double d = e10 < 0 ? (double)m / P10[-e10] : (double)m * P10[e10];
                uint64_t bits;
                memcpy(&bits, &d, 8);
                const uint64_t frac = bits & 0x1FFFFFFFull;
                const bool     safe = m == 0 || ((d > 1.0e-30 && d < 1.0e30) && (frac < 0x0FFFFFFFull || frac > 0x10000001ull));
                if (safe) {
                    out[i] = neg ? -(float)d : (float)d;
                    p      = q;
                    continue;
                }
            }
        }
        char* end = NULL;
        out[i]    = strtof(p, &end);
        if (end != q) return false;
        p = q;
    }
    return true;
}
static bool fast_corner(const char* t, ssize_t* vi, ssize_t* vt, ssize_t* vn) // "a", "a/b", "a//c", "a/b/c", plain digits
{
    size_t f[3]   = {0, 0, 0};
    bool   has[3] = {false, false, false};
    int    k      = 0;
    for (;;) {
        int digits = 0;
        while (*t >= '0' && *t <= '9') {
            if (++digits > 18) return false;
            f[k] = f[k] * 10 + (size_t)(*t - '0');
            t++;
        }
        has[k] = digits > 0;
        if (*t == '/' && k < 2) {
            k++;
            t++;
            continue;
        }
        break;
    }
    if (!(*t == 0 || *t == '\n')) return false; // "\r", spaces, letters: sscanf decides
    if (!has[0]) return false;
    if (k == 0) {
        *vi = (ssize_t)f[0];
    } else if (k == 1) {
        if (!has[1]) return false;
        *vi = (ssize_t)f[0], *vt = (ssize_t)f[1];
    } else {
        if (!has[2]) return false; // "a/b/" and "a//": sscanf decides
        *vi = (ssize_t)f[0], *vn = (ssize_t)f[2];
        if (has[1]) *vt = (ssize_t)f[1];
    }
    return true;
}

// The importer's two loop bodies (mesh.c:104-127 and :129-243), one fgets chunk at a time, shared by the FILE*
// entry point (sequential, real fgets) and the path entry point (the file in memory, chunks parsed by several
// threads).  Errors come back as text: the workers are not the thread whose last-error slot the caller reads.
struct ObjVerts {
    std::vector<point3> v, vn;
    std::vector<point2> vt;
};

static void obj_vertex_chunk(const char* lineBuffer, ObjVerts& out) // first pass: v / vt / vn
{
    point3 vn, v;
    point2 vt;
    float  fv[3];
    if (lineBuffer[0] != 'v') {
        // none of the three patterns can match a line that does not start with 'v'
    } else if (is_blank(lineBuffer[1]) && fast_floats(lineBuffer + 1, 3, fv)) {
        out.v.push_back(point3{fv[0], fv[1], fv[2]});
    } else if (lineBuffer[1] == 'n' && is_blank(lineBuffer[2]) && fast_floats(lineBuffer + 2, 3, fv)) {
        out.vn.push_back(point3{fv[0], fv[1], fv[2]});
    } else if (lineBuffer[1] == 't' && is_blank(lineBuffer[2]) && fast_floats(lineBuffer + 2, 2, fv)) {
        out.vt.push_back(point2{fv[0], fv[1]});
    } else if (sscanf(lineBuffer, "v %f %f %f\n", &v.x, &v.y, &v.z) == 3) {
        out.v.push_back(v);
    } else if (sscanf(lineBuffer, "vt %f %f\n", &vt.x, &vt.y) == 2) {
        out.vt.push_back(vt);
    } else if (sscanf(lineBuffer, "vn %f %f %f\n", &vn.x, &vn.y, &vn.z) == 3) {
        out.vn.push_back(vn);
    }
}

static std::string fmt_err(const char* fmt, const char* arg = "")
{
    char b[600];
    snprintf(b, sizeof b, fmt, arg);
    return b;
}

// second pass: one chunk that starts with "f " (the caller checked); the chunk is chopped by strtok like the
// reference's line buffer.  false + *err on what the reference ABORTs or overruns on.
static bool obj_face_chunk(char* lineBuffer, const ObjVerts& g, TriangleVec& polys, std::string* err)
{
    const std::vector<point3>&vertices = g.v, &vertexNormals = g.vn;
    const std::vector<point2>& texCoords = g.vt;

    ssize_t vi[4] = {-1, -1, -1, -1}, vn[4] = {-1, -1, -1, -1}, vt[4] = {-1, -1, -1, -1};
    size_t  tc;
    char*   save  = NULL; // strtok_r: same tokens as strtok, usable from several threads
    char*   token = strtok_r(lineBuffer, " ", &save);
    token         = strtok_r(NULL, " ", &save);
    for (tc = 0; token != NULL; tc++) {
        if (tc >= 4) {
            *err = fmt_err("Mesh_Import_OBJ: face with more than 4 vertices (the reference overruns vi[4], mesh.c:136)");
            return false;
        }
        if (fast_corner(token, &vi[tc], &vt[tc], &vn[tc])) {
        } else if (sscanf(token, "%zu/%zu/%zu", (size_t*)&vi[tc], (size_t*)&vt[tc], (size_t*)&vn[tc]) == 3) {
        } else if (sscanf(token, "%zu//%zu", (size_t*)&vi[tc], (size_t*)&vn[tc]) == 2) {
        } else if (sscanf(token, "%zu/%zu", (size_t*)&vi[tc], (size_t*)&vt[tc]) == 2) {
        } else if (sscanf(token, "%zu", (size_t*)&vi[tc]) == 1) {
        } else {
            *err = fmt_err("Mesh_Import_OBJ: no match for token \"%s\" (the reference ABORTs, mesh.c:155)", token);
            return false;
        }
        token = strtok_r(NULL, " ", &save);
    }

    auto corner_ok = [&](size_t c) {
        return vi[c] >= 1 && (size_t)vi[c] <= vertices.size() && (vn[c] < 0 || (vn[c] >= 1 && (size_t)vn[c] <= vertexNormals.size()))
               && (vt[c] < 0 || (vt[c] >= 1 && (size_t)vt[c] <= texCoords.size()));
    };
    for (size_t c = 0; c < tc; c++) {
        if (!corner_ok(c)) {
            *err = fmt_err("Mesh_Import_OBJ: face index out of range");
            return false;
        }
    }

    if (tc == 4) { // quad → (0,1,2),(0,2,3); default normal NOT normalised (mesh.c:179-181)
        static const size_t indices[2][3] = {{0, 1, 2}, {0, 2, 3}};
        for (size_t ii = 0; ii < 2; ii++) {
            Triangle tri;
            for (size_t jj = 0; jj < 3; jj++) tri.vtx[jj].pos = vertices[vi[indices[ii][jj]] - 1];
            vec3 defaultNormal = vcross(vsub(tri.vtx[1].pos, tri.vtx[0].pos), vsub(tri.vtx[2].pos, tri.vtx[0].pos));
            for (size_t jj = 0; jj < 3; jj++) {
                size_t index      = indices[ii][jj];
                tri.vtx[jj].norm = vn[index] >= 0 ? vertexNormals[vn[index] - 1] : defaultNormal;
                tri.vtx[jj].tex  = vt[index] >= 0 ? texCoords[vt[index] - 1] : point2{0.0f, 0.0f};
            }
            polys.push_back(tri);
        }
    } else if (tc == 3) { // triangle; default normal normalised (mesh.c:212-214)
        Triangle tri;
        for (size_t ii = 0; ii < 3; ii++) tri.vtx[ii].pos = vertices[vi[ii] - 1];
        vec3 defaultNormal = vnorm(vcross(vsub(tri.vtx[1].pos, tri.vtx[0].pos), vsub(tri.vtx[2].pos, tri.vtx[0].pos)));
        for (size_t ii = 0; ii < 3; ii++) {
            tri.vtx[ii].norm = vn[ii] >= 0 ? vertexNormals[vn[ii] - 1] : defaultNormal;
            tri.vtx[ii].tex  = vt[ii] >= 0 ? texCoords[vt[ii] - 1] : point2{0.0f, 0.0f};
        }
        polys.push_back(tri);
    }
    return true;
}

extern "C" bool Mesh_Import_OBJ(Mesh* mesh, FILE* fd) // mesh.c:87-244
{
    char     lineBuffer[512];
    ObjVerts g;

    lineBuffer[0] = 0;
    // first pass: v / vt / vn.  `while(!feof) fgets` re-examines the last line once more when the
    // file ends in a newline, exactly as the reference does (a trailing "v" line is pushed twice;
    // harmless because nothing indexes the duplicate).
    while (!feof(fd)) {
        if (!fgets(lineBuffer, sizeof lineBuffer, fd)) { /* buffer unchanged, like the reference */ }
        obj_vertex_chunk(lineBuffer, g);
    }

    fseek(fd, 0, SEEK_SET);
    lineBuffer[0] = 0;

    while (!feof(fd)) {
        if (!fgets(lineBuffer, sizeof lineBuffer, fd)) {}
        if (strncmp(lineBuffer, "f ", 2) != 0) continue;
        std::string err;
        if (!obj_face_chunk(lineBuffer, g, mesh->polys, &err)) {
            set_err("%s", err.c_str());
            return false;
        }
    }
    return true;
}

// fgets(buf, 512, fd) on a file held in memory: up to 511 bytes, through the first '\n'.  The caller has checked
// p < end.  *eof is what feof() would say afterwards: set only when the read ran into the end of the file (not
// when it stopped at a newline or on a full buffer that happens to be the file's last byte).
static const char* mem_fgets(char* buf, const char* p, const char* end, bool* eof)
{
    size_t room = 511, n = 0;
    while (n < room && p + n < end) {
        const char c = p[n];
        buf[n++]     = c;
        if (c == '\n') {
            buf[n] = 0;
            *eof   = false;
            return p + n;
        }
    }
    buf[n] = 0;
    *eof   = n < room; // stopped because the file ended, not because the buffer was full
    return p + n;
}

static unsigned import_threads(size_t bytes)
{
    // a small file is parsed before the threads would have started (unless the caller insists)
    return (bytes < (1u << 20) && !getenv("RTX_HOST_THREADS")) ? 1 : host_threads();
}

// The importer over a file in memory, by several threads (on a 1.1 M-triangle file parsing is
// what is left of the scene build once the kd-tree is built on the device).  The file is cut at physical line
// starts, so every thread sees the fgets chunks the sequential loop would see (a line longer than 511 bytes is
// several chunks, each examined as a line of its own — also here); vertices and triangles are concatenated in
// file order; the one chunk the reference's `while(!feof)` loop examines twice — the last, when the file does not
// end inside it — is examined twice here too.  Result and error behaviour are those of Mesh_Import_OBJ
// (`tests/test_host_build.py` compares the two on every asset and on the loader's corner cases).
static bool import_obj_memory(Mesh* mesh, const char* data, size_t size)
{
    // more blocks than threads, handed out on demand: an OBJ file is all vertices, then all faces, so equal byte
    // ranges are very unequal work in either pass
    const unsigned      T = import_threads(size);
    const unsigned      B = T == 1 ? 1 : T * 8;
    std::vector<size_t> cut(B + 1, size);
    cut[0] = 0;
    for (unsigned k = 1; k < B; k++) {
        size_t      at = std::max(cut[k - 1], size / B * k);
        const void* nl = at < size ? memchr(data + at, '\n', size - at) : NULL;
        cut[k]         = nl ? (size_t)((const char*)nl - data) + 1 : size;
    }
    struct Part {
        ObjVerts              verts;
        std::vector<size_t>   faces; // offsets of the chunks that start with "f "
        TriangleVec           polys;
        std::string           err;
        bool                  ok = true;
    };
    std::vector<Part> part(B);
    char              lastChunk[512];
    bool              lastEof = true; // feof() after the chunk that reaches the end of the file
    lastChunk[0]              = 0;

    auto run = [&](auto&& body) { // body(block) for every block, T threads
        std::atomic<unsigned>    next(0);
        auto                     worker = [&] { for (unsigned b; (b = next.fetch_add(1)) < B;) body(b); };
        std::vector<std::thread> th;
        for (unsigned k = 1; k < T; k++) th.emplace_back(worker);
        worker();
        for (auto& t : th) t.join();
    };
    const bool   timing = getenv("RTX_HOST_TIMING") != NULL;
    auto         now    = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
    const double t0     = now();

    run([&](unsigned k) { // first pass
        char        buf[512];
        const char *p = data + cut[k], *end = data + cut[k + 1];
        bool        eof = false;
        Part&       me  = part[k];
        while (p < end) {
            const char* at = p;
            p              = mem_fgets(buf, p, data + size, &eof);
            obj_vertex_chunk(buf, me.verts);
            if (buf[0] == 'f' && buf[1] == ' ') me.faces.push_back((size_t)(at - data));
            if (p >= data + size) { // only the block that owns the end of the file gets here
                memcpy(lastChunk, buf, sizeof buf);
                lastEof = eof;
            }
        }
    });

    const double        t1 = now();
    ObjVerts            g;
    std::vector<size_t> faces;
    {
        size_t nv = 0, nn = 0, nt = 0, nf = 0;
        for (auto& q : part) nv += q.verts.v.size(), nn += q.verts.vn.size(), nt += q.verts.vt.size(), nf += q.faces.size();
        g.v.reserve(nv + 1), g.vn.reserve(nn + 1), g.vt.reserve(nt + 1), faces.reserve(nf);
        for (auto& q : part) {
            g.v.insert(g.v.end(), q.verts.v.begin(), q.verts.v.end());
            g.vn.insert(g.vn.end(), q.verts.vn.begin(), q.verts.vn.end());
            g.vt.insert(g.vt.end(), q.verts.vt.begin(), q.verts.vt.end());
            faces.insert(faces.end(), q.faces.begin(), q.faces.end());
            q.verts = ObjVerts();
            std::vector<size_t>().swap(q.faces);
        }
    }
    if (size > 0 && !lastEof) obj_vertex_chunk(lastChunk, g); // the loop's extra turn: fgets fails, the buffer is examined again
    // (in the second pass that extra turn sees a face line already chopped to "f\0…" by strtok, or a line that
    // is no face: nothing happens)

    const double t2 = now();
    run([&](unsigned k) { // second pass: the face chunks, in file order, in B equal runs
        char         buf[512];
        bool         eof = false;
        Part&        me  = part[k];
        const size_t lo = faces.size() * k / B, hi = faces.size() * (k + 1) / B;
        me.polys.reserve((hi - lo) + (hi - lo) / 8);
        for (size_t i = lo; i < hi; i++) {
            mem_fgets(buf, data + faces[i], data + size, &eof);
            if (!obj_face_chunk(buf, g, me.polys, &me.err)) {
                me.ok = false;
                return;
            }
        }
    });

    const double t3 = now();
    // triangles in file order, up to and including the run that failed (what the sequential loop leaves behind)
    std::vector<size_t> base(B + 1, mesh->polys.size());
    unsigned            used = 0;
    for (unsigned k = 0; k < B; k++) {
        base[k + 1] = base[k] + part[k].polys.size();
        used        = k + 1;
        if (!part[k].ok) break;
    }
    if (B == 1 && mesh->polys.empty()) {
        mesh->polys.swap(part[0].polys);
    } else {
        mesh->polys.resize(base[used]);
        run([&](unsigned k) {
            if (k < used && !part[k].polys.empty()) memcpy(&mesh->polys[base[k]], part[k].polys.data(), part[k].polys.size() * sizeof(Triangle));
        });
    }
    if (timing) fprintf(stderr, "import_obj_memory: %u threads, vertices %.3f s, concat %.3f s, faces %.3f s, triangles %.3f s\n", T, t1 - t0, t2 - t1, t3 - t2, now() - t3);
    if (!part[used - 1].ok) {
        set_err("%s", part[used - 1].err.c_str());
        return false;
    }
    return true;
}

extern "C" bool Mesh_Import_OBJ_Memory(Mesh* mesh, const char* data, size_t size)
{
    if (!mesh || (!data && size)) {
        set_err("Mesh_Import_OBJ_Memory: null argument");
        return false;
    }
    return import_obj_memory(mesh, data, size);
}

extern "C" bool Mesh_Import_OBJ_Path(Mesh* mesh, const char* path)
{
    FILE* fd = fopen(path, "r");
    if (!fd) {
        set_err("cannot open %s", path);
        return false;
    }
    std::vector<char> data;
    size_t            have = 0, got;
    if (fseek(fd, 0, SEEK_END) == 0) {
        const long n = ftell(fd);
        if (n > 0) data.resize((size_t)n);
        fseek(fd, 0, SEEK_SET);
    }
    for (;;) { // the size is a hint: read until fread says there is no more
        if (have == data.size()) data.resize(have + (1u << 16));
        got = fread(data.data() + have, 1, data.size() - have, fd);
        if (got == 0) break;
        have += got;
    }
    data.resize(have);
    const bool bad = ferror(fd) != 0;
    fclose(fd);
    if (bad) {
        set_err("cannot read %s", path);
        return false;
    }
    return import_obj_memory(mesh, data.data(), data.size());
}

// ---------------------------------------------------------------------------------------------
// camera (src/world/camera.c:10-32)
// ---------------------------------------------------------------------------------------------
static_assert(sizeof(Camera) == 88 && sizeof(rtx_camera) == 88, "Camera block is 88 bytes");

extern "C" Camera* Camera_New(point3 lookFrom, point3 lookTo, vec3 vup, float aspectRatio, float vertFov,
                              float aperature, float focusDist)
{
    float theta          = vertFov * PI32 / 180.0f; // radiansf, math.c:20-23
    float hh             = tanf(theta / 2.0f);
    float viewportHeight = 2.0f * hh;
    float viewportWidth  = aspectRatio * viewportHeight;

    Camera* cam = (Camera*)calloc(1, sizeof *cam);
    if (!cam) return NULL;
    cam->w = vnorm(vsub(lookFrom, lookTo));
    cam->u = vnorm(vcross(vup, cam->w));
    cam->v = vcross(cam->w, cam->u);

    cam->origin     = lookFrom;
    cam->horizontal = vmuls(cam->u, viewportWidth * focusDist);
    cam->vertical   = vmuls(cam->v, viewportHeight * focusDist);
    // vsum(a,b,c) = (a+b)+c  (vec.h:389-390)
    vec3 s                = vadd(vadd(vmuls(cam->horizontal, 0.5f), vmuls(cam->vertical, 0.5f)), vmuls(cam->w, focusDist));
    cam->bottomLeftCorner = vsub(cam->origin, s);
    cam->lensRadius       = aperature / 2.0f;
    return cam;
}

extern "C" void Camera_Delete(Camera* cam) { free(cam); }

// ---------------------------------------------------------------------------------------------
// flattener: Scene → rtx_scene_desc pools
// ---------------------------------------------------------------------------------------------
struct SceneFlat {
    rtx_scene_desc                desc;
    std::vector<uint32_t>         leaf_ref;
    std::vector<float>            tri_pos, tri_nrm, tri_uv, sph, plane, tex_data;
    std::vector<uint8_t>          tri_mat, sph_mat, plane_mat;
    std::vector<uint32_t>         tri_obj, sph_obj;
    std::vector<rtx_material>     materials;
    std::vector<rtx_texture_desc> textures;
    std::vector<uint32_t>         w0, w1;
    std::unordered_map<const Material*, int32_t> matIndex;
};

extern "C" SceneFlat* Scene_Flatten(Scene* scene)
{
    if (!scene || !scene->kdTree) {
        set_err("Scene_Flatten: scene has no kd-tree (call Scene_Prepare; at least one bounded object is required)");
        return NULL;
    }
    SceneFlat* f = new SceneFlat();
    KDTree*    t = scene->kdTree;

    std::unordered_map<const Texture*, int32_t>   texIndex;
    std::unordered_map<const Material*, int32_t>& matIndex = f->matIndex;

    auto intern_tex = [&](const Texture* tex) -> int32_t {
        if (!tex || !tex->image) return -1;
        auto it = texIndex.find(tex);
        if (it != texIndex.end()) return it->second;
        rtx_texture_desc d;
        d.width  = (uint32_t)tex->image->res.width;
        d.height = (uint32_t)tex->image->res.height;
        d.offset = f->tex_data.size() / 3;
        const float* src = (const float*)tex->image->pix;
        f->tex_data.insert(f->tex_data.end(), src, src + 3 * (size_t)d.width * d.height);
        int32_t id = (int32_t)f->textures.size();
        f->textures.push_back(d);
        texIndex[tex] = id;
        return id;
    };

    int32_t skyFaces[6] = {-1, -1, -1, -1, -1, -1};
    bool    skyDone     = false;
    auto    intern_sky  = [&](const Skybox* sky) {
        if (!sky || skyDone) return;
        for (int i = 0; i < 6; i++) skyFaces[i] = intern_tex(sky->tex[i]);
        skyDone = true;
    };
    intern_sky(scene->skybox);

    bool ok         = true;
    auto intern_mat = [&](const Material* m) -> uint8_t {
        if (!m) {
            ok = false;
            set_err("Scene_Flatten: object without material");
            return 0;
        }
        auto it = matIndex.find(m);
        if (it != matIndex.end()) return (uint8_t)it->second;
        if (f->materials.size() >= 256) {
            ok = false;
            set_err("Scene_Flatten: more than 256 distinct materials");
            return 0;
        }
        rtx_material r;
        memset(&r, 0, sizeof r);
        r.type = m->type;
        memcpy(r.p, m->p, sizeof m->p);
        if (m->type == RTX_MAT_SKYBOX) {
            // the reference's Material_Skybox points at a Skybox; one skybox per scene is supported
            if (m->skybox != scene->skybox) {
                ok = false;
                set_err("Scene_Flatten: Material_Skybox must reference the scene's own skybox");
            }
            r.tex = -1;
        } else {
            r.tex = intern_tex(m->albedo);
        }
        int32_t id = (int32_t)f->materials.size();
        f->materials.push_back(r);
        matIndex[m] = id;
        return (uint8_t)id;
    };

    // bounded primitives, in kdObjects order
    const size_t nk = scene->kdSeq.size();
    size_t       T = 0, S = 0;
    for (size_t i = 0; i < nk; i++) ((scene->kdSeq[i] & SCENE_OTHER) ? S : T)++;
    std::vector<uint32_t> kd2ref(nk);
    f->tri_pos.resize(9 * T);
    f->tri_nrm.resize(9 * T);
    f->tri_uv.resize(6 * T);
    f->tri_mat.resize(T);
    f->tri_obj.resize(T);
    f->sph.resize(4 * S);
    f->sph_mat.resize(S);
    f->sph_obj.resize(S);
    // slots and materials in kdObjects order (cheap, sequential: the order materials are first met fixes their
    // indices), then the transposition of the triangle pool into the SoA planes on all cores
    size_t          ti = 0, si = 0;
    const Material* lastMat = NULL;
    uint8_t         lastId  = 0;
    for (size_t i = 0; i < nk; i++) {
        const uint32_t e = scene->kdSeq[i];
        if (!(e & SCENE_OTHER)) {
            const Material* m = scene->triMat[e];
            if (m != lastMat || !m) lastId = intern_mat(m), lastMat = m;
            f->tri_mat[ti] = lastId;
            f->tri_obj[ti] = (uint32_t)i;
            kd2ref[i]      = (uint32_t)ti;
            ti++;
        } else {
            const Object& o = scene->others[e & ~SCENE_OTHER];
            f->sph[0 * S + si] = o.surface.sphere.c.x;
            f->sph[1 * S + si] = o.surface.sphere.c.y;
            f->sph[2 * S + si] = o.surface.sphere.c.z;
            f->sph[3 * S + si] = o.surface.sphere.r;
            f->sph_mat[si]     = intern_mat(o.material);
            f->sph_obj[si]     = (uint32_t)i;
            kd2ref[i]          = RTX_REF_SPHERE | (uint32_t)si;
            si++;
        }
    }
    parallel_ranges(T, [&](size_t lo, size_t hi) {
        for (size_t t = lo; t < hi; t++) {
            const Triangle& tr = scene->tris[scene->kdSeq[f->tri_obj[t]]];
            for (int v = 0; v < 3; v++) {
                f->tri_pos[(3 * v + 0) * T + t] = tr.vtx[v].pos.x;
                f->tri_pos[(3 * v + 1) * T + t] = tr.vtx[v].pos.y;
                f->tri_pos[(3 * v + 2) * T + t] = tr.vtx[v].pos.z;
                f->tri_nrm[(3 * v + 0) * T + t] = tr.vtx[v].norm.x;
                f->tri_nrm[(3 * v + 1) * T + t] = tr.vtx[v].norm.y;
                f->tri_nrm[(3 * v + 2) * T + t] = tr.vtx[v].norm.z;
                f->tri_uv[(2 * v + 0) * T + t]  = tr.vtx[v].tex.x;
                f->tri_uv[(2 * v + 1) * T + t]  = tr.vtx[v].tex.y;
            }
        }
    });
    const size_t P = scene->unboundSeq.size();
    f->plane.resize(6 * P);
    f->plane_mat.resize(P);
    for (size_t i = 0; i < P; i++) {
        const Object& o      = scene->others[scene->unboundSeq[i]];
        f->plane[0 * P + i] = o.surface.plane.normal.x;
        f->plane[1 * P + i] = o.surface.plane.normal.y;
        f->plane[2 * P + i] = o.surface.plane.normal.z;
        f->plane[3 * P + i] = o.surface.plane.point.x;
        f->plane[4 * P + i] = o.surface.plane.point.y;
        f->plane[5 * P + i] = o.surface.plane.point.z;
        f->plane_mat[i]     = intern_mat(o.material);
    }
    if (!ok) {
        delete f;
        return NULL;
    }

    f->w0 = t->w0;
    f->w1 = t->w1;
    f->leaf_ref.resize(t->refs.size());
    for (size_t i = 0; i < t->refs.size(); i++) f->leaf_ref[i] = kd2ref[t->refs[i]];

    rtx_scene_desc& d = f->desc;
    memset(&d, 0, sizeof d);
    d.abi_version = RTX_ABI_VERSION;
    d.n_nodes     = (uint32_t)f->w0.size();
    d.n_refs      = (uint32_t)f->leaf_ref.size();
    d.root_index  = (uint32_t)t->rootIndex;
    d.n_tris      = (uint32_t)T;
    d.n_spheres   = (uint32_t)S;
    d.n_planes    = (uint32_t)P;
    d.n_materials = (uint32_t)f->materials.size();
    d.n_textures  = (uint32_t)f->textures.size();
    d.max_depth   = (uint32_t)t->maxDepth + 1;
    d.node_w0     = f->w0.data();
    d.node_w1     = f->w1.data();
    d.leaf_ref    = f->leaf_ref.data();
    d.tri_pos     = f->tri_pos.data();
    d.tri_nrm     = f->tri_nrm.data();
    d.tri_uv      = f->tri_uv.data();
    d.tri_mat     = f->tri_mat.data();
    d.tri_obj     = f->tri_obj.data();
    d.sph         = f->sph.data();
    d.sph_mat     = f->sph_mat.data();
    d.sph_obj     = f->sph_obj.data();
    d.plane       = f->plane.data();
    d.plane_mat   = f->plane_mat.data();
    d.materials   = f->materials.data();
    d.textures    = f->textures.data();
    d.tex_data    = f->tex_data.data();
    d.tex_texels  = f->tex_data.size() / 3;
    for (int i = 0; i < 6; i++) d.skybox_tex[i] = skyFaces[i];
    return f;
}

extern "C" const rtx_scene_desc* SceneFlat_Desc(const SceneFlat* flat) { return flat ? &flat->desc : NULL; }
extern "C" void                  SceneFlat_Delete(SceneFlat* flat) { delete flat; }

extern "C" int SceneFlat_MaterialIndex(const SceneFlat* flat, const Material* material)
{
    auto it = flat->matIndex.find(material);
    return it == flat->matIndex.end() ? -1 : it->second;
}

// ---------------------------------------------------------------------------------------------
// render driver (src/rt/renderer.c:23-52, 239-274): same asynchronous contract, GPU behind it
// ---------------------------------------------------------------------------------------------
struct HipLib {
    void* handle;
    int (*scene_create)(const rtx_scene_desc*, int, rtx_scene**);
    int (*scene_destroy)(rtx_scene*);
    int (*render)(rtx_scene*, const rtx_camera*, const rtx_render_params*, uint8_t*, float*, rtx_stats*);
    int (*render_progressive)(rtx_scene*, const rtx_camera*, const rtx_render_params*, uint32_t, uint8_t*, float*, rtx_progress_fn, void*, rtx_stats*);
    int (*render_multi)(rtx_scene* const*, uint32_t, const rtx_camera*, const rtx_render_params*, uint32_t, uint32_t, uint8_t*, float*, rtx_stats*);
    const char* (*last_error)(void);
};

static HipLib* load_hip_lib()
{
    static HipLib lib;
    static bool   tried = false;
    if (tried) return lib.handle ? &lib : NULL;
    tried = true;
    std::string path;
    if (const char* env = getenv("RTX_HIP_LIB")) {
        path = env;
    } else {
        Dl_info info;
        if (dladdr((void*)&load_hip_lib, &info) && info.dli_fname) {
            path       = info.dli_fname;
            size_t pos = path.find_last_of('/');
            path       = (pos == std::string::npos ? std::string(".") : path.substr(0, pos)) + "/librt_hip.so";
        } else {
            path = "librt_hip.so";
        }
    }
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        set_err("Render_Start: cannot load %s: %s", path.c_str(), dlerror());
        return NULL;
    }
    lib.scene_create  = (decltype(lib.scene_create))dlsym(h, "rtx_scene_create");
    lib.scene_destroy = (decltype(lib.scene_destroy))dlsym(h, "rtx_scene_destroy");
    lib.render        = (decltype(lib.render))dlsym(h, "rtx_render");
    lib.last_error    = (decltype(lib.last_error))dlsym(h, "rtx_last_error");
    lib.render_progressive = (decltype(lib.render_progressive))dlsym(h, "rtx_render_progressive");
    lib.render_multi       = (decltype(lib.render_multi))dlsym(h, "rtx_render_multi");
    if (!lib.scene_create || !lib.scene_destroy || !lib.render || !lib.render_progressive || !lib.render_multi || !lib.last_error) {
        set_err("Render_Start: %s lacks the rtx_* entry points", path.c_str());
        dlclose(h);
        return NULL;
    }
    lib.handle = h;
    return &lib;
}

struct WaiterArgs {
    RenderCtx* ctx;
    size_t     samples_per_pixel, max_ray_depth;
    uint64_t   seed_lo, seed_hi;
    int        device;
    int        devices[64]; // Render_Set_Devices: the devices that share the frame (n_devices > 1: rtx_render_multi)
    uint32_t   n_devices, split_mode, band_rows;
    uint32_t   spp_per_pass; // 0: the frame appears at once; else it fills in by passes of that many samples
    volatile int stop;       // Render_Stop: end the frame after the pass in flight
    volatile uint32_t samples_done;
    int        status;
    rtx_stats  stats;
    char       err[512];
    pthread_t  thread;
    bool       thread_live;
};

extern "C" RenderCtx* Render_New(Scene* scene, ImageRGB* img, Camera* cam)
{
    RenderCtx*  ctx = (RenderCtx*)calloc(1, sizeof *ctx);
    WaiterArgs* wa  = (WaiterArgs*)calloc(1, sizeof *wa);
    if (!ctx || !wa) {
        free(ctx);
        free(wa);
        return NULL;
    }
    wa->ctx            = ctx;
    wa->seed_lo        = 1;
    wa->seed_hi        = 2;
    ctx->cam           = cam;
    ctx->scene         = scene;
    ctx->img           = img;
    ctx->waiter_thread = NULL;
    ctx->waiter_args   = wa;
    ctx->finished      = true;
    return ctx;
}

static void join_waiter(WaiterArgs* wa)
{
    if (wa->thread_live) {
        pthread_join(wa->thread, NULL);
        wa->thread_live = false;
    }
}

extern "C" void Render_Delete(RenderCtx* ctx)
{
    if (!ctx) return;
    WaiterArgs* wa = (WaiterArgs*)ctx->waiter_args;
    join_waiter(wa);
    free(wa);
    free(ctx);
}

extern "C" void Render_Set_Seed(RenderCtx* ctx, uint64_t lo, uint64_t hi)
{
    WaiterArgs* wa = (WaiterArgs*)ctx->waiter_args;
    wa->seed_lo    = lo;
    wa->seed_hi    = hi;
}

extern "C" void Render_Set_Device(RenderCtx* ctx, int device)
{
    WaiterArgs* wa = (WaiterArgs*)ctx->waiter_args;
    wa->device     = device;
    wa->n_devices  = 0;
}

// The frame of the next Render_Start is shared by n devices, as Render_Waiter shares it among NUM_HYPERTHREADS workers
// (renderer.c:186-237): split_mode RTX_SPLIT_TILE (interleaved bands of band_rows rows, 0 = 8) or RTX_SPLIT_SAMPLE.
// Ignored (false) for n > 64; a progressive frame (Render_Set_Progressive) stays on devices[0].
extern "C" bool Render_Set_Devices(RenderCtx* ctx, const int* devices, uint32_t n, uint32_t split_mode, uint32_t band_rows)
{
    WaiterArgs* wa = (WaiterArgs*)ctx->waiter_args;
    if (!devices || n == 0 || n > 64) return false;
    memcpy(wa->devices, devices, n * sizeof(int));
    wa->device = devices[0], wa->n_devices = n, wa->split_mode = split_mode, wa->band_rows = band_rows;
    return true;
}

// The reference's image fills in tile by tile while main.c blits it every 200 ms (main.c:436-478); here it
// fills in by samples: after every pass of `spp_per_pass` samples ctx->img->pix holds the image so far.
extern "C" void Render_Set_Progressive(RenderCtx* ctx, uint32_t spp_per_pass) { ((WaiterArgs*)ctx->waiter_args)->spp_per_pass = spp_per_pass; }
// main.c:47-58 (SIGINT: stop and export what is there): ends a progressive frame after the pass in flight;
// Render_Done then turns true with the partial image in ctx->img->pix.
extern "C" void     Render_Stop(RenderCtx* ctx) { ((WaiterArgs*)ctx->waiter_args)->stop = 1; }
extern "C" uint32_t Render_Samples_Done(RenderCtx* ctx) { return ((WaiterArgs*)ctx->waiter_args)->samples_done; }

static int Render_Progress(void* user, uint32_t done, uint32_t)
{
    WaiterArgs* wa   = (WaiterArgs*)user;
    wa->samples_done = done;
    return wa->stop;
}

extern "C" int Render_Status(RenderCtx* ctx, rtx_stats* stats)
{
    WaiterArgs* wa = (WaiterArgs*)ctx->waiter_args;
    if (stats) *stats = wa->stats;
    if (wa->status != RTX_OK) set_err("%s", wa->err);
    return wa->status;
}

static void* Render_Waiter(void* arg) // stands in for Render_Waiter + Render_Worker, renderer.c:141-237
{
    WaiterArgs* wa  = (WaiterArgs*)arg;
    RenderCtx*  ctx = wa->ctx;
    wa->status      = RTX_E_INVALID;
    wa->err[0]      = 0;
    HipLib*    lib  = load_hip_lib();
    SceneFlat* flat = NULL;
    rtx_scene* dev  = NULL;
    if (!lib) {
        snprintf(wa->err, sizeof wa->err, "%s", g_err);
        wa->status = RTX_E_NO_DEVICE;
        goto done;
    }
    flat = Scene_Flatten(ctx->scene);
    if (!flat) {
        snprintf(wa->err, sizeof wa->err, "%s", g_err);
        goto done;
    }
    if (wa->n_devices > 1 && !wa->spp_per_pass) { // one frame over several devices: the library runs one worker per device
        rtx_scene* devs[64] = {NULL};
        uint32_t   made     = 0;
        wa->status          = RTX_OK;
        for (; made < wa->n_devices && wa->status == RTX_OK; made++) wa->status = lib->scene_create(&flat->desc, wa->devices[made], &devs[made]);
        if (wa->status != RTX_OK) {
            snprintf(wa->err, sizeof wa->err, "%s", lib->last_error());
            made--; // the last one was not created
        } else {
            rtx_render_params p;
            memset(&p, 0, sizeof p);
            p.width = (uint32_t)ctx->img->res.width, p.height = (uint32_t)ctx->img->res.height;
            p.spp = (uint32_t)wa->samples_per_pixel, p.max_depth = (uint32_t)wa->max_ray_depth;
            p.seed_lo = wa->seed_lo, p.seed_hi = wa->seed_hi;
            std::vector<rtx_stats> st(wa->n_devices);
            wa->status = lib->render_multi(devs, wa->n_devices, (const rtx_camera*)ctx->cam, &p, wa->split_mode, wa->band_rows, (uint8_t*)ctx->img->pix, NULL, st.data());
            if (wa->status != RTX_OK) snprintf(wa->err, sizeof wa->err, "%s", lib->last_error());
            else { // the job's stats: counts summed over the devices, times of the slowest
                wa->stats = st[0];
                for (uint32_t r = 1; r < wa->n_devices; r++) {
                    wa->stats.rays_total += st[r].rays_total, wa->stats.camera_paths += st[r].camera_paths, wa->stats.hits_shaded += st[r].hits_shaded;
                    wa->stats.rays_early_miss += st[r].rays_early_miss, wa->stats.trace_launches += st[r].trace_launches, wa->stats.passes += st[r].passes;
                    for (int k = 0; k < RTX_MAX_DEPTH; k++) wa->stats.rays_at_depth[k] += st[r].rays_at_depth[k];
                    wa->stats.ms_total = std::max(wa->stats.ms_total, st[r].ms_total), wa->stats.ms_trace = std::max(wa->stats.ms_trace, st[r].ms_trace);
                    wa->stats.ms_shade = std::max(wa->stats.ms_shade, st[r].ms_shade);
                }
                wa->stats.samples_done = p.spp;
                wa->samples_done       = p.spp;
            }
        }
        for (uint32_t r = 0; r < made; r++) lib->scene_destroy(devs[r]);
        goto done;
    }
    wa->status = lib->scene_create(&flat->desc, wa->device, &dev);
    if (wa->status != RTX_OK) {
        snprintf(wa->err, sizeof wa->err, "%s", lib->last_error());
        goto done;
    }
    {
        rtx_render_params p;
        memset(&p, 0, sizeof p);
        p.width     = (uint32_t)ctx->img->res.width;
        p.height    = (uint32_t)ctx->img->res.height;
        p.spp       = (uint32_t)wa->samples_per_pixel;
        p.max_depth = (uint32_t)wa->max_ray_depth;
        p.seed_lo   = wa->seed_lo;
        p.seed_hi   = wa->seed_hi;
        static_assert(sizeof(RGB) == 3, "RGB is 3 bytes");
        if (wa->spp_per_pass)
            wa->status = lib->render_progressive(dev, (const rtx_camera*)ctx->cam, &p, wa->spp_per_pass, (uint8_t*)ctx->img->pix, NULL, Render_Progress,
                                                 wa, &wa->stats);
        else
            wa->status = lib->render(dev, (const rtx_camera*)ctx->cam, &p, (uint8_t*)ctx->img->pix, NULL, &wa->stats);
        wa->samples_done = wa->stats.samples_done;
        if (wa->status != RTX_OK) snprintf(wa->err, sizeof wa->err, "%s", lib->last_error());
    }
done:
    if (dev) lib->scene_destroy(dev);
    if (flat) SceneFlat_Delete(flat);
    __atomic_store_n(&ctx->finished, true, __ATOMIC_RELEASE);
    return NULL;
}

extern "C" void Render_Start(RenderCtx* ctx, size_t samples_per_pixel, size_t max_ray_depth)
{
    WaiterArgs* wa = (WaiterArgs*)ctx->waiter_args;
    join_waiter(wa);
    ctx->finished         = false;
    wa->stop              = 0;
    wa->samples_done      = 0;
    wa->samples_per_pixel = samples_per_pixel;
    wa->max_ray_depth     = max_ray_depth;
    if (pthread_create(&wa->thread, NULL, Render_Waiter, wa) != 0) {
        wa->status = RTX_E_NOMEM;
        snprintf(wa->err, sizeof wa->err, "Render_Start: cannot start the waiter thread");
        ctx->finished = true;
        return;
    }
    wa->thread_live    = true;
    ctx->waiter_thread = &wa->thread;
}

extern "C" bool Render_Done(RenderCtx* ctx) { return __atomic_load_n(&ctx->finished, __ATOMIC_ACQUIRE); }
