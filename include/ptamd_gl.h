/* ptamd_gl.h — optional OpenGL presenter for the surface ptamd_raytrace() writes (libptamd_gl.so).
 *
 * Replaces, for an application that HAS a window, the GL half of the reference's driver::Interop
 * (cuda_opengl/include/driver/interop.h:18-80, src/driver/interop.cpp):
 *   interop.cpp:14-20,104-116  two renderbuffers + framebuffers, cudaGraphicsGLRegisterImage   -> create / resize
 *   interop.cpp:36-56          cudaGraphicsMapResources / SubResourceGetMappedArray / Unmap     -> inside present()
 *   interop.cpp:67-72          glBlitNamedFramebuffer(..., 0, height, width, 0, ...) (flipped)  -> present()
 * The megakernel writes a LINEAR RGBA8 device buffer (row 0 = top), not a cudaArray, so the interop object here is a
 * pixel-unpack buffer registered with HIP (hipGraphicsGLRegisterBuffer): present() maps it, copies the surface into it
 * device-to-device on the caller's stream, unmaps, uploads it into a texture (glTexSubImage2D from the bound PBO: no
 * host round trip) and blits the texture's framebuffer to the default framebuffer, flipped like the reference's blit
 * (GL's origin is bottom-left).
 *
 * The CALLER owns the window and the GL context (GLFW in the reference: main.cpp:120-166) and makes it current on the
 * calling thread; this library creates no window.  An MI355X node has no display, so this part of the product is
 * compile- and link-tested only (tests/test_abi.py); without a current GL 3.0+ context create() fails with
 * PTAMD_ERR_ARG and a message, it never falls back to anything.
 */
#ifndef PTAMD_GL_H
#define PTAMD_GL_H

#include "ptamd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ptamd_gl_presenter ptamd_gl_presenter;

/* Message of the last failed call of this library on the calling thread (status codes are ptamd.h's). */
const char* ptamd_gl_get_last_error(void);

/* Needs a current OpenGL >= 3.0 context on this thread and the HIP device that drives it. */
int ptamd_gl_presenter_create(uint32_t width, uint32_t height, ptamd_gl_presenter** out);
/* Interop::setSize (interop.cpp:82-117): re-creates texture + buffer at the new size. */
int ptamd_gl_presenter_resize(ptamd_gl_presenter* p, uint32_t width, uint32_t height);
/* surface_rgba8: device pointer, width*height*4 bytes, row 0 = top (what ptamd_raytrace wrote); stream: hipStream_t the
 * render was issued on (the copy is ordered behind it).  Draws into the default framebuffer; the caller swaps buffers
 * (glfwSwapBuffers, main.cpp:200). */
int ptamd_gl_presenter_present(ptamd_gl_presenter* p, const void* surface_rgba8, void* stream);
void ptamd_gl_presenter_destroy(ptamd_gl_presenter* p);

#ifdef __cplusplus
}
#endif
#endif
