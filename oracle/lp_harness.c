/*
 * lp_harness.c -- headless OpenGL 3.3 core context on Mesa llvmpipe, no X / GLX / EGL.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is part of the oracle: it exists so that the
 * reference renderer's own GLSL (read as text at run time from /root/reference, never
 * copied into this repository) can be executed on the CPU to produce golden vectors.
 * Nothing in the shipped renderer links, loads or calls it.
 *
 * It plays the role of the reference's window/context + shader helpers
 * (main.py:134-161 context creation, main.py:67-85 / main.py:195-208 compile+link,
 * main.py:212-220 full-screen quad, main.py:277-300 uniform upload + draw) with the
 * window replaced by an off-screen framebuffer object that is read back.
 *
 * How: dlopen libglapi + Mesa's swrast_dri.so, talk to the DRI "swrast" extension
 * directly with stub loader callbacks, create a core-profile context, render the
 * 6-vertex quad into an RGBA8 (product-equivalent) or RGBA32F (unclamped, for
 * instrumentation) renderbuffer and glReadPixels it (rows bottom-up, as GL defines).
 *
 * Built by oracle/Makefile into oracle/_ref/liblp_harness.so and driven through ctypes
 * by oracle/llvmpipe.py.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>

static void *g_glapi, *g_drv;
static const __DRIcoreExtension *g_core;
static const __DRIswrastExtension *g_swrast;
static __DRIscreen *g_screen;
static __DRIcontext *g_ctx;
static __DRIdrawable *g_draw;
static int g_w = 64, g_h = 64;
static char g_err[4096];

static GLuint g_prog, g_vao, g_vbo, g_fbo, g_rbo;
static int g_fbo_w, g_fbo_h, g_fbo_float = -1;

/* ---- stub loader callbacks: the drawable is never presented anywhere ---- */
static void cb_get_drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p)
{ (void)d; (void)p; *x = 0; *y = 0; *w = g_w; *h = g_h; }
static void cb_put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void cb_get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void cb_put_image2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void cb_get_image2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }

static const __DRIswrastLoaderExtension g_loader = {
    .base = { __DRI_SWRAST_LOADER, 3 },
    .getDrawableInfo = cb_get_drawable_info,
    .putImage = cb_put_image,
    .getImage = cb_get_image,
    .putImage2 = cb_put_image2,
    .getImage2 = cb_get_image2,
};
static const __DRIextension *g_loader_exts[] = { &g_loader.base, NULL };

/* ---- GL entry points, resolved through glapi ---- */
#define GLFN(ret, name, ...) static ret (APIENTRY *p_##name)(__VA_ARGS__)
GLFN(GLuint, glCreateShader, GLenum);
GLFN(void, glShaderSource, GLuint, GLsizei, const GLchar *const *, const GLint *);
GLFN(void, glCompileShader, GLuint);
GLFN(void, glGetShaderiv, GLuint, GLenum, GLint *);
GLFN(void, glGetShaderInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLFN(GLuint, glCreateProgram, void);
GLFN(void, glAttachShader, GLuint, GLuint);
GLFN(void, glLinkProgram, GLuint);
GLFN(void, glGetProgramiv, GLuint, GLenum, GLint *);
GLFN(void, glGetProgramInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLFN(void, glUseProgram, GLuint);
GLFN(void, glDeleteShader, GLuint);
GLFN(void, glDeleteProgram, GLuint);
GLFN(GLint, glGetUniformLocation, GLuint, const GLchar *);
GLFN(void, glUniform1i, GLint, GLint);
GLFN(void, glUniform1fv, GLint, GLsizei, const GLfloat *);
GLFN(void, glUniform2fv, GLint, GLsizei, const GLfloat *);
GLFN(void, glUniform3fv, GLint, GLsizei, const GLfloat *);
GLFN(void, glUniform4fv, GLint, GLsizei, const GLfloat *);
GLFN(void, glUniformMatrix4fv, GLint, GLsizei, GLboolean, const GLfloat *);
GLFN(void, glGenVertexArrays, GLsizei, GLuint *);
GLFN(void, glBindVertexArray, GLuint);
GLFN(void, glGenBuffers, GLsizei, GLuint *);
GLFN(void, glBindBuffer, GLenum, GLuint);
GLFN(void, glBufferData, GLenum, GLsizeiptr, const void *, GLenum);
GLFN(void, glVertexAttribPointer, GLuint, GLint, GLenum, GLboolean, GLsizei, const void *);
GLFN(void, glEnableVertexAttribArray, GLuint);
GLFN(void, glGenFramebuffers, GLsizei, GLuint *);
GLFN(void, glBindFramebuffer, GLenum, GLuint);
GLFN(void, glDeleteFramebuffers, GLsizei, const GLuint *);
GLFN(void, glGenRenderbuffers, GLsizei, GLuint *);
GLFN(void, glBindRenderbuffer, GLenum, GLuint);
GLFN(void, glDeleteRenderbuffers, GLsizei, const GLuint *);
GLFN(void, glRenderbufferStorage, GLenum, GLenum, GLsizei, GLsizei);
GLFN(void, glFramebufferRenderbuffer, GLenum, GLenum, GLenum, GLuint);
GLFN(GLenum, glCheckFramebufferStatus, GLenum);
GLFN(void, glViewport, GLint, GLint, GLsizei, GLsizei);
GLFN(void, glClearColor, GLfloat, GLfloat, GLfloat, GLfloat);
GLFN(void, glClear, GLbitfield);
GLFN(void, glDrawArrays, GLenum, GLint, GLsizei);
GLFN(void, glFinish, void);
GLFN(void, glReadPixels, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void *);
GLFN(void, glPixelStorei, GLenum, GLint);
GLFN(const GLubyte *, glGetString, GLenum);
GLFN(GLenum, glGetError, void);
GLFN(void, glGetIntegerv, GLenum, GLint *);

static void *(*p_get_proc)(const char *);

static int resolve_all(void)
{
#define R(name) do { *(void **)&p_##name = p_get_proc(#name); \
        if (!p_##name) { snprintf(g_err, sizeof g_err, "missing GL entry %s", #name); return -1; } } while (0)
    R(glCreateShader); R(glShaderSource); R(glCompileShader); R(glGetShaderiv); R(glGetShaderInfoLog);
    R(glCreateProgram); R(glAttachShader); R(glLinkProgram); R(glGetProgramiv); R(glGetProgramInfoLog);
    R(glUseProgram); R(glDeleteShader); R(glDeleteProgram); R(glGetUniformLocation);
    R(glUniform1i); R(glUniform1fv); R(glUniform2fv); R(glUniform3fv); R(glUniform4fv); R(glUniformMatrix4fv);
    R(glGenVertexArrays); R(glBindVertexArray); R(glGenBuffers); R(glBindBuffer); R(glBufferData);
    R(glVertexAttribPointer); R(glEnableVertexAttribArray);
    R(glGenFramebuffers); R(glBindFramebuffer); R(glDeleteFramebuffers);
    R(glGenRenderbuffers); R(glBindRenderbuffer); R(glDeleteRenderbuffers);
    R(glRenderbufferStorage); R(glFramebufferRenderbuffer); R(glCheckFramebufferStatus);
    R(glViewport); R(glClearColor); R(glClear); R(glDrawArrays); R(glFinish); R(glReadPixels);
    R(glPixelStorei); R(glGetString); R(glGetError); R(glGetIntegerv);
#undef R
    return 0;
}

const char *lp_last_error(void) { return g_err; }

/* Create the context.  Returns 0 on success. */
int lp_open(void)
{
    if (g_ctx) return 0;
    g_glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!g_glapi) { snprintf(g_err, sizeof g_err, "dlopen libglapi: %s", dlerror()); return -1; }
    const char *paths[] = {
        "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so",
        "/usr/lib64/dri/swrast_dri.so",
        "/usr/lib/dri/swrast_dri.so",
        "swrast_dri.so", NULL };
    for (int i = 0; paths[i] && !g_drv; ++i)
        g_drv = dlopen(paths[i], RTLD_NOW | RTLD_GLOBAL);
    if (!g_drv) { snprintf(g_err, sizeof g_err, "dlopen swrast_dri.so: %s", dlerror()); return -1; }

    const __DRIextension **(*get_exts)(void) =
        (const __DRIextension **(*)(void))dlsym(g_drv, __DRI_DRIVER_GET_EXTENSIONS "_swrast");
    if (!get_exts) { snprintf(g_err, sizeof g_err, "driver has no swrast extension getter"); return -1; }
    const __DRIextension **drv_exts = get_exts();
    for (int i = 0; drv_exts[i]; ++i) {
        if (!strcmp(drv_exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension *)drv_exts[i];
        if (!strcmp(drv_exts[i]->name, __DRI_SWRAST)) g_swrast = (const __DRIswrastExtension *)drv_exts[i];
    }
    if (!g_core || !g_swrast || g_swrast->base.version < 4) {
        snprintf(g_err, sizeof g_err, "DRI_Core / DRI_SWRast(v4) not offered by the driver"); return -1;
    }
    const __DRIconfig **configs = NULL;
    g_screen = g_swrast->createNewScreen2(0, g_loader_exts, drv_exts, &configs, NULL);
    if (!g_screen || !configs || !configs[0]) { snprintf(g_err, sizeof g_err, "createNewScreen2 failed"); return -1; }
    const __DRIconfig *cfg = configs[0];
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 3 };
    unsigned err = 0;
    g_ctx = g_swrast->createContextAttribs(g_screen, __DRI_API_OPENGL_CORE, cfg, NULL, 2, attribs, &err, NULL);
    if (!g_ctx) { snprintf(g_err, sizeof g_err, "createContextAttribs failed (dri error %u)", err); return -1; }
    g_draw = g_swrast->createNewDrawable(g_screen, cfg, NULL);
    if (!g_draw) { snprintf(g_err, sizeof g_err, "createNewDrawable failed"); return -1; }
    if (!g_core->bindContext(g_ctx, g_draw, g_draw)) { snprintf(g_err, sizeof g_err, "bindContext failed"); return -1; }

    p_get_proc = (void *(*)(const char *))dlsym(g_glapi, "_glapi_get_proc_address");
    if (!p_get_proc) { snprintf(g_err, sizeof g_err, "_glapi_get_proc_address missing"); return -1; }
    if (resolve_all()) return -1;

    /* the reference's full-screen quad, main.py:212 */
    static const float quad[12] = { -1, 1, -1, -1, 1, -1, -1, 1, 1, -1, 1, 1 };
    p_glGenVertexArrays(1, &g_vao);
    p_glGenBuffers(1, &g_vbo);
    p_glBindVertexArray(g_vao);
    p_glBindBuffer(GL_ARRAY_BUFFER, g_vbo);
    p_glBufferData(GL_ARRAY_BUFFER, sizeof quad, quad, GL_STATIC_DRAW);
    p_glVertexAttribPointer(0, 2, GL_FLOAT, GL_FALSE, 8, (void *)0);
    p_glEnableVertexAttribArray(0);
    return 0;
}

/* which: 0 renderer, 1 version, 2 GLSL version, 3 vendor */
const char *lp_gl_string(int which)
{
    static const GLenum e[4] = { GL_RENDERER, GL_VERSION, GL_SHADING_LANGUAGE_VERSION, GL_VENDOR };
    if (!g_ctx || which < 0 || which > 3) return "";
    return (const char *)p_glGetString(e[which]);
}

int lp_get_integer(unsigned pname)
{
    GLint v = -1;
    if (g_ctx) p_glGetIntegerv(pname, &v);
    return v;
}

static GLuint compile(GLenum type, const char *src)
{
    GLuint s = p_glCreateShader(type);
    p_glShaderSource(s, 1, &src, NULL);
    p_glCompileShader(s);
    GLint ok = 0;
    p_glGetShaderiv(s, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[3500] = "";
        p_glGetShaderInfoLog(s, sizeof log, NULL, log);
        snprintf(g_err, sizeof g_err, "compile error (%s): %s", type == GL_VERTEX_SHADER ? "vs" : "fs", log);
        p_glDeleteShader(s);
        return 0;
    }
    return s;
}

/* Compile + link + use a program from source text held in memory. */
int lp_program(const char *vs_src, const char *fs_src)
{
    if (!g_ctx) { snprintf(g_err, sizeof g_err, "lp_open not called"); return -1; }
    GLuint vs = compile(GL_VERTEX_SHADER, vs_src);
    if (!vs) return -1;
    GLuint fs = compile(GL_FRAGMENT_SHADER, fs_src);
    if (!fs) { p_glDeleteShader(vs); return -1; }
    GLuint prog = p_glCreateProgram();
    p_glAttachShader(prog, vs);
    p_glAttachShader(prog, fs);
    p_glLinkProgram(prog);
    p_glDeleteShader(vs);
    p_glDeleteShader(fs);
    GLint ok = 0;
    p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[3500] = "";
        p_glGetProgramInfoLog(prog, sizeof log, NULL, log);
        snprintf(g_err, sizeof g_err, "link error: %s", log);
        p_glDeleteProgram(prog);
        return -1;
    }
    if (g_prog) p_glDeleteProgram(g_prog);
    g_prog = prog;
    p_glUseProgram(g_prog);
    return 0;
}

/* Uniform setters: return 0 if set, 1 if the name is not an active uniform. */
int lp_uniform_i(const char *name, int v)
{
    GLint loc = p_glGetUniformLocation(g_prog, name);
    if (loc < 0) return 1;
    p_glUniform1i(loc, v);
    return 0;
}

int lp_uniform_f(const char *name, int n, const float *v)
{
    GLint loc = p_glGetUniformLocation(g_prog, name);
    if (loc < 0) return 1;
    switch (n) {
    case 1: p_glUniform1fv(loc, 1, v); break;
    case 2: p_glUniform2fv(loc, 1, v); break;
    case 3: p_glUniform3fv(loc, 1, v); break;
    case 4: p_glUniform4fv(loc, 1, v); break;
    default: return -1;
    }
    return 0;
}

int lp_uniform_m4(const char *name, int transpose, const float *m)
{
    GLint loc = p_glGetUniformLocation(g_prog, name);
    if (loc < 0) return 1;
    p_glUniformMatrix4fv(loc, 1, transpose ? GL_TRUE : GL_FALSE, m);
    return 0;
}

static int ensure_target(int w, int h, int is_float)
{
    if (g_fbo && g_fbo_w == w && g_fbo_h == h && g_fbo_float == is_float) return 0;
    if (g_fbo) { p_glDeleteFramebuffers(1, &g_fbo); p_glDeleteRenderbuffers(1, &g_rbo); g_fbo = g_rbo = 0; }
    p_glGenFramebuffers(1, &g_fbo);
    p_glGenRenderbuffers(1, &g_rbo);
    p_glBindRenderbuffer(GL_RENDERBUFFER, g_rbo);
    p_glRenderbufferStorage(GL_RENDERBUFFER, is_float ? GL_RGBA32F : GL_RGBA8, w, h);
    p_glBindFramebuffer(GL_FRAMEBUFFER, g_fbo);
    p_glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_RENDERBUFFER, g_rbo);
    GLenum st = p_glCheckFramebufferStatus(GL_FRAMEBUFFER);
    if (st != GL_FRAMEBUFFER_COMPLETE) {
        snprintf(g_err, sizeof g_err, "framebuffer incomplete: 0x%x", st);
        return -1;
    }
    g_fbo_w = w; g_fbo_h = h; g_fbo_float = is_float;
    return 0;
}

/*
 * Draw the quad into a w x h target and read it back.
 *   is_float = 0: RGBA8 UNORM target, out = w*h*4 bytes
 *   is_float = 1: RGBA32F target,     out = w*h*4 floats
 * Rows are bottom-up (GL window coordinates).  Returns 0 on success.
 */
int lp_draw(int w, int h, int is_float, void *out)
{
    if (!g_prog) { snprintf(g_err, sizeof g_err, "no program"); return -1; }
    if (ensure_target(w, h, is_float)) return -1;
    p_glBindFramebuffer(GL_FRAMEBUFFER, g_fbo);
    p_glViewport(0, 0, w, h);
    p_glClearColor(0.f, 0.f, 0.f, 1.f);
    p_glClear(GL_COLOR_BUFFER_BIT);
    p_glUseProgram(g_prog);
    p_glBindVertexArray(g_vao);
    p_glDrawArrays(GL_TRIANGLES, 0, 6);
    p_glFinish();
    if (out) {
        p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
        p_glReadPixels(0, 0, w, h, GL_RGBA, is_float ? GL_FLOAT : GL_UNSIGNED_BYTE, out);
    }
    GLenum e = p_glGetError();
    if (e != GL_NO_ERROR) { snprintf(g_err, sizeof g_err, "GL error 0x%x", e); return -1; }
    return 0;
}
