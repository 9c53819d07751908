// probe: dlopen the engine, create and destroy a context (no solve)
#include <cstdio>
#include <dlfcn.h>
#include <cstdlib>
struct psd_ctx;
int main(int argc, char** argv) {
    void* h = dlopen(getenv("LIBPSD_MI355X"), RTLD_NOW);
    if (!h) { printf("dlopen failed %s\n", dlerror()); return 2; }
    auto create = (int (*)(psd_ctx**, int))dlsym(h, "psd_create");
    auto destroy = (int (*)(psd_ctx*))dlsym(h, "psd_destroy");
    psd_ctx* c = nullptr;
    int rc = create(&c, 0);
    printf("create rc=%d\n", rc);
    destroy(c);
    return 0;
}
