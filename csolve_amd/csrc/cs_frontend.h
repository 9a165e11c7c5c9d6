/* cs_frontend.h -- hand-written recursive-descent front end for the csolve
 * problem text format.
 *
 * The reference generates its front end with flex/bison (reference
 * src/lexer.l:36-102, src/parser.y:94-267); neither tool exists in this image,
 * and the generated sources are not part of the reference tree.  This parser
 * accepts the same token set and the same grammar and reports every reduction
 * to a *builder* (a table of callbacks), in the order the reference's semantic
 * actions run.  Two builders exist:
 *   - csolve_amd/csrc/cs_pool.c   builds the product's index-based node pool;
 *   - oracle/ref_harness.c        builds the reference's own pointer trees
 *                                 (only in this container, for golden vectors).
 */
#ifndef CS_FRONTEND_H
#define CS_FRONTEND_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* operator codes shared by the flat model (cs_flat.h) and the builders */
enum cs_op {
  CS_OP_VAR = 0,   /* terminal bound to a variable: a = variable index          */
  CS_OP_CONST = 1, /* terminal without variable: a = lo, b = hi                 */
  CS_OP_EQ = 2,
  CS_OP_LT = 3,
  CS_OP_NEG = 4,
  CS_OP_ADD = 5,
  CS_OP_MUL = 6,
  CS_OP_NOT = 7,
  CS_OP_AND = 8,
  CS_OP_OR = 9,
  CS_OP_WAND = 10, /* n-ary "wide and": a = offset into kids[], b = count       */
  CS_OP_CONFL = 11 /* learnt conflict clause (not produced by the front end)    */
};

/* objective kinds, numbering of reference csolve.h:242-247 */
enum cs_objective { CS_OBJ_ANY = 0, CS_OBJ_ALL = 1, CS_OBJ_MIN = 2, CS_OBJ_MAX = 3 };

/* initial-order weights, reference src/parser_support.h:23-27 */
#define CS_WEIGHT_EQUAL 1000
#define CS_WEIGHT_COMPARE 100
#define CS_WEIGHT_NOT_EQUAL 10

/* Builder callbacks.  Expression handles are opaque to the parser. */
typedef struct cs_builder {
  void *ctx;
  /* NUM -> constant terminal (parser.y:135-138) */
  void *(*num)(void *ctx, int32_t value);
  /* IDENT -> the variable's terminal, created [-inf,+inf] on first mention (139-148) */
  void *(*ident)(void *ctx, const char *name);
  /* NEG / NOT (154-161) */
  void *(*unary)(void *ctx, int op, void *child);
  /* EQ LT ADD MUL AND OR (192-283) */
  void *(*binary)(void *ctx, int op, void *l, void *r);
  /* nested wide-and over already built elements (all_different, 163-184) */
  void *(*wand)(void *ctx, void **elems, size_t n);
  /* vars_weighten(expr, weight / max(1, vars_count(expr))) when weights are on (219-265) */
  void (*weigh)(void *ctx, void *expr, int32_t weight);
  /* objective statement; expr is NULL for ANY/ALL.  Returns the first root element (109-131) */
  void *(*objective)(void *ctx, int kind, void *expr);
  /* append one element to the root wide-and (94-106) */
  void (*constraint)(void *ctx, void *expr);
} cs_builder;

/* Parse `text` (NUL-terminated).  Returns 0 on success; on a lexical or syntax
 * error returns -1 and writes a message "... in line N" to err (the reference
 * prints "invalid input `c' in line N" / "<bison message> in line N" and exits,
 * lexer.l:95-98, parser.y:288-290). */
int cs_parse_text(const char *text, const cs_builder *b, char *err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
