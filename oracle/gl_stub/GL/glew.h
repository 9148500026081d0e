// OpenGL type/constant stub — TEST INFRASTRUCTURE.  GLEW/freeglut are not installed in this image; the reference's
// anim/Character.h pulls render/DrawMesh.h (and with it <GL/glew.h>) into the kinematic-character sources that
// oracle/_ref compiles.  Only the type names and the few enum constants those headers mention are declared; no GL
// function exists and none is called by anything oracle/_ref exercises (draw is disabled).
#pragma once
typedef unsigned int GLuint;
typedef unsigned int GLenum;
typedef int GLint;
typedef int GLsizei;
typedef float GLfloat;
typedef double GLdouble;
typedef unsigned char GLubyte;
typedef unsigned char GLboolean;
typedef void GLvoid;
typedef char GLchar;
typedef unsigned int GLbitfield;
typedef long GLsizeiptr;
typedef long GLintptr;
#define GL_ARRAY_BUFFER 0x8892
#define GL_ELEMENT_ARRAY_BUFFER 0x8893
#define GL_RGBA 0x1908
#define GL_RGBA8 0x8058
#define GL_RGBA32F 0x8814
#define GL_TRIANGLES 0x0004
#define GL_LINES 0x0001
#define GL_POINTS 0x0000
#define GL_FLOAT 0x1406
#define GL_UNSIGNED_BYTE 0x1401
#define GL_UNSIGNED_INT 0x1405
#define GL_FALSE 0
#define GL_TRUE 1
#define GL_STATIC_DRAW 0x88E4
// Entry points named by inline code in render/RenderState.h.  They are declared so that the headers parse; the
// bodies abort: nothing oracle/_ref runs may reach the renderer.
#include <cstdlib>
#define DM_GL_STUB(name) template <class... A> inline void name(A...) { std::abort(); }
DM_GL_STUB(glBindBuffer)
DM_GL_STUB(glBindVertexArray)
DM_GL_STUB(glBufferData)
DM_GL_STUB(glBufferSubData)
DM_GL_STUB(glDeleteBuffers)
DM_GL_STUB(glDeleteVertexArrays)
DM_GL_STUB(glEnableVertexAttribArray)
DM_GL_STUB(glGenBuffers)
DM_GL_STUB(glGenVertexArrays)
DM_GL_STUB(glVertexAttribPointer)
#undef DM_GL_STUB
