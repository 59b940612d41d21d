// LD_PRELOAD shim: live bytes / counts per allocation size class (exact usable size for sizes >= 4 KB kept in a small table)
#define _GNU_SOURCE
#include <dlfcn.h>
#include <malloc.h>
#include <stdatomic.h>
#include <stdio.h>
#include <string.h>
static void* (*real_malloc)(size_t) = 0;
static void (*real_free)(void*) = 0;
static void* (*real_calloc)(size_t, size_t) = 0;
static void* (*real_realloc)(void*, size_t) = 0;
static void* (*real_memalign)(size_t, size_t) = 0;
static _Atomic long cnt[64], byt[64];
static char boot[65536]; static size_t boot_used = 0;
static int bucket(size_t s) { int b = 0; while (s > 1 && b < 63) { s >>= 1; ++b; } return b; }
static void init(void) {
  real_malloc = dlsym(RTLD_NEXT, "malloc"); real_free = dlsym(RTLD_NEXT, "free");
  real_calloc = dlsym(RTLD_NEXT, "calloc"); real_realloc = dlsym(RTLD_NEXT, "realloc"); real_memalign = dlsym(RTLD_NEXT, "memalign");
}
static void add(void* p) { if (!p) return; size_t s = malloc_usable_size(p); int b = bucket(s); cnt[b]++; byt[b] += (long)s; }
static void sub(void* p) { if (!p) return; size_t s = malloc_usable_size(p); int b = bucket(s); cnt[b]--; byt[b] -= (long)s; }
void* malloc(size_t n) { if (!real_malloc) init(); void* p = real_malloc(n); add(p); return p; }
void free(void* p) { if (!p) return; if ((char*)p >= boot && (char*)p < boot + sizeof boot) return; if (!real_free) init(); sub(p); real_free(p); }
void* calloc(size_t a, size_t b) {
  if (!real_calloc) { size_t n = a * b; void* p = boot + boot_used; boot_used += (n + 15) & ~15ul; memset(p, 0, n); return p; }
  void* p = real_calloc(a, b); add(p); return p; }
void* realloc(void* p, size_t n) { if (!real_realloc) init(); if ((char*)p >= boot && (char*)p < boot + sizeof boot) { void* q = real_malloc(n); add(q); return q; } sub(p); void* q = real_realloc(p, n); add(q); return q; }
void* memalign(size_t al, size_t n) { if (!real_memalign) init(); void* p = real_memalign(al, n); add(p); return p; }
int posix_memalign(void** out, size_t al, size_t n) { void* p = memalign(al, n); *out = p; return p ? 0 : 12; }
void* aligned_alloc(size_t al, size_t n) { return memalign(al, n); }
void mshim_dump(long* c, long* b) { for (int i = 0; i < 64; ++i) { c[i] = cnt[i]; b[i] = byt[i]; } }
