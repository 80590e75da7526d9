/*
 * ref_prog_shim.c -- TEST INFRASTRUCTURE ONLY.  Two link-time wraps for the UNMODIFIED reference program
 * (MIMC_main.c + MIMC_module.c + GMA.c + MIMC_misc.c, compiled where they lie by `make -C oracle refprog`
 * into oracle/_ref/MIMC3_ref) so that a run can be repeated:
 *   malloc -> zero-filling (T4: the program reads memory it never wrote; fresh large mallocs are zero pages)
 *   time   -> $MIMC3_REF_SEED when set (the CP stage seeds its shuffle with time(NULL), MIMC_module.c:517)
 */
#include <stdlib.h>
#include <time.h>

void *__wrap_malloc(size_t n) { return calloc(1, n ? n : 1); }

time_t __real_time(time_t *t);
time_t __wrap_time(time_t *t)
{
    const char *e = getenv("MIMC3_REF_SEED");
    if (!e) return __real_time(t);
    const time_t v = (time_t)atol(e);
    if (t) *t = v;
    return v;
}
