/* Test-only helpers: comparators and input generators used by the known-answer tests.
 * They re-create the INPUTS of the reference's example programs (never their code):
 *  - kat_alphacmp: case-insensitive wide-char comparator, the behaviour of
 *    examples/aho_corasick_generic_test.c:48-54.
 *  - kat_rand_letters: the unseeded glibc rand() letter stream of generic_test.c:263-273.
 *  - kat_read_novel: the fgetwc/iswalpha/towlower normalisation of generic_test.c:191-195.
 */
#define _GNU_SOURCE
#include <locale.h>
#include <stdio.h>
#include <stdlib.h>
#include <wchar.h>
#include <wctype.h>

int
kat_alphacmp (const void *a, const void *b, const void *arg) {
  (void)arg;
  wint_t x = towlower (*(const wint_t *)a), y = towlower (*(const wint_t *)b);
  return x > y ? 1 : (x < y ? -1 : 0);
}

/* next n letters 'a' + rand() % 26, continuing the process-wide rand() stream */
void
kat_rand_letters (unsigned char *out, size_t n) {
  for (size_t i = 0; i < n; i++)
    out[i] = (unsigned char)('a' + (size_t)rand () % 26);
}

void
kat_srand (unsigned seed) {
  srand (seed);
}

/* Reads a UTF-8 file wide-char by wide-char; non-alphabetic -> L' ', alphabetic -> lower case.
 * Returns the number of wide chars written (<= cap), or -1 on error. */
long
kat_read_novel (const char *path, wchar_t *out, long cap) {
  if (!setlocale (LC_ALL, "C.UTF-8") && !setlocale (LC_ALL, "C.utf8"))
    return -1;
  FILE *f = fopen (path, "r");
  if (!f)
    return -1;
  long n = 0;
  for (wint_t wc; (wc = fgetwc (f)) != WEOF && n < cap;)
    out[n++] = iswalpha (wc) ? (wchar_t)towlower (wc) : L' ';
  fclose (f);
  return n;
}

/* comparators for the comparator-class path (1- and 2-byte symbols): the byte / UTF-16-unit
 * analogues of kat_alphacmp, and one whose classes interleave over the value range */
int
kat_casecmp8 (const void *a, const void *b, const void *arg) {
  (void)arg;
  unsigned x = *(const unsigned char *)a, y = *(const unsigned char *)b;
  if (x >= 'A' && x <= 'Z')
    x += 32;
  if (y >= 'A' && y <= 'Z')
    y += 32;
  return x > y ? 1 : (x < y ? -1 : 0);
}

int
kat_casecmp16 (const void *a, const void *b, const void *arg) {
  (void)arg;
  wint_t x = towlower ((wint_t) * (const unsigned short *)a), y = towlower ((wint_t) * (const unsigned short *)b);
  return x > y ? 1 : (x < y ? -1 : 0);
}

int
kat_mod7cmp8 (const void *a, const void *b, const void *arg) {
  (void)arg;
  unsigned x = *(const unsigned char *)a % 7u, y = *(const unsigned char *)b % 7u;
  return x > y ? 1 : (x < y ? -1 : 0);
}

/* not an order: a < b < c < a */
int
kat_cyclic_cmp8 (const void *a, const void *b, const void *arg) {
  (void)arg;
  unsigned x = *(const unsigned char *)a % 3u, y = *(const unsigned char *)b % 3u;
  if (x == y)
    return 0;
  return (x + 1) % 3 == y ? -1 : 1;
}

/* the reference's alphacmp (generic_test.c:48-54) as it stands: case-insensitive order of wchar_t
 * symbols (4 bytes here) through towlower */
int
kat_casecmp32 (const void *a, const void *b, const void *arg) {
  (void)arg;
  wint_t x = towlower (*(const wint_t *)a), y = towlower (*(const wint_t *)b);
  return x > y ? 1 : (x < y ? -1 : 0);
}

/* not an order over 4-byte symbols: a < b < c < a */
int
kat_cyclic_cmp32 (const void *a, const void *b, const void *arg) {
  (void)arg;
  unsigned x = *(const unsigned *)a % 3u, y = *(const unsigned *)b % 3u;
  if (x == y)
    return 0;
  return (x + 1) % 3 == y ? -1 : 1;
}

/* towlower beyond ASCII needs a UTF-8 locale (the reference's test calls setlocale (LC_ALL, "")) */
#include <locale.h>
void
kat_setlocale_utf8 (void) {
  if (!setlocale (LC_CTYPE, "C.UTF-8"))
    setlocale (LC_CTYPE, "C.utf8");
}
