/* ref_gl_boot.c — TEST INFRASTRUCTURE.  Brings up a real OpenGL 4.6 core context on this GPU-less machine: Mesa's llvmpipe,
 * loaded straight from /usr/lib/x86_64-linux-gnu/dri/swrast_dri.so through the DRI software-rasteriser loader interface
 * (GL/internal/dri_interface.h) — no X server, no EGL, no OSMesa needed.  GL entry points are then resolved through
 * libglapi's _glapi_get_proc_address.  Used by ref_gl_check.cpp to run the reference's UNMODIFIED conversion shaders
 * through a real GL implementation (rasteriser, interpolation, mip generation, texture filtering, atomic counter). */
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void get_drawable_info(__DRIdrawable* d, int* x, int* y, int* w, int* h, void* p) { (void)d; (void)p; *x = 0; *y = 0; *w = 16; *h = 16; }
static void put_image(__DRIdrawable* d, int op, int x, int y, int w, int h, char* data, void* p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void get_image(__DRIdrawable* d, int x, int y, int w, int h, char* data, void* p) { (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * (size_t)h * 4); }
static void put_image2(__DRIdrawable* d, int op, int x, int y, int w, int h, int s, char* data, void* p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)s; (void)data; (void)p; }

typedef void* (*gl_get_proc_fn)(const char*);

/* returns the entry-point resolver, or NULL (message in *why) */
gl_get_proc_fn ref_gl_boot(const char** why) {
    static __DRIswrastLoaderExtension loader;
    static const __DRIextension* loader_exts[2];
    const char* paths[] = { "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", "swrast_dri.so", NULL };
    void* h = NULL;
    setenv("MESA_GL_VERSION_OVERRIDE", "4.6", 0);
    setenv("MESA_GLSL_VERSION_OVERRIDE", "460", 0);
    for (int i = 0; paths[i] && !h; ++i) h = dlopen(paths[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) { *why = "swrast_dri.so (Mesa llvmpipe) could not be loaded"; return NULL; }
    const __DRIextension** (*getext)(void) = (const __DRIextension** (*)(void))dlsym(h, "__driDriverGetExtensions_swrast");
    if (!getext) { *why = "swrast_dri.so lacks __driDriverGetExtensions_swrast"; return NULL; }
    const __DRIextension** exts = getext();
    const __DRIcoreExtension* core = NULL;
    const __DRIswrastExtension* swrast = NULL;
    for (int i = 0; exts[i]; ++i) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension*)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension*)exts[i];
    }
    if (!core || !swrast || swrast->base.version < 4) { *why = "DRI core / swrast (v4) extension missing"; return NULL; }
    memset(&loader, 0, sizeof loader);
    loader.base.name = __DRI_SWRAST_LOADER; loader.base.version = 2;
    loader.getDrawableInfo = get_drawable_info; loader.putImage = put_image; loader.getImage = get_image; loader.putImage2 = put_image2;
    loader_exts[0] = &loader.base; loader_exts[1] = NULL;
    const __DRIconfig** configs = NULL;
    __DRIscreen* screen = swrast->createNewScreen2(0, loader_exts, exts, &configs, NULL);
    if (!screen || !configs || !configs[0]) { *why = "createNewScreen2 failed"; return NULL; }
    unsigned err = 0;
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5 };   /* glewGlfwHandler.cpp:14-16 asks for 4.5 core */
    __DRIcontext* ctx = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!ctx) { *why = "createContextAttribs(OpenGL 4.5 core) failed"; return NULL; }
    __DRIdrawable* draw = swrast->createNewDrawable(screen, configs[0], NULL);
    if (!draw || !core->bindContext(ctx, draw, draw)) { *why = "bindContext failed"; return NULL; }
    void* glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) { *why = "libglapi.so.0 could not be loaded"; return NULL; }
    gl_get_proc_fn gpa = (gl_get_proc_fn)dlsym(glapi, "_glapi_get_proc_address");
    if (!gpa) { *why = "libglapi lacks _glapi_get_proc_address"; return NULL; }
    return gpa;
}
