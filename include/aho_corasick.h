/* Source-compatibility shim: programs written against farhiongit/aho-corasick-1975 include
 * "aho_corasick.h"; the declarations live in acm.h. */
#include "acm.h"
