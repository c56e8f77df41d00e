/* TEST INFRASTRUCTURE.  The reference's main() seeds every random choice with time(NULL)
 * (/root/reference/src/main.cpp:37,40,301,303,...), so two runs never see the same
 * inputs.  Linked into the test builds of the reference's program, this definition takes
 * precedence over libc's for calls made from the executable and makes the run reproducible:
 * the same binary sources over the plaintext-bit provider (CPU, golden output) and over
 * libtfhe-hip (GPU) must then print the same lines. */
#include <time.h>
time_t time(time_t *t) {
    if (t) *t = (time_t)1700000000;
    return (time_t)1700000000;
}
