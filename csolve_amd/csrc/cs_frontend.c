/* cs_frontend.c -- recursive-descent parser for csolve problem text.
 * Token set: reference src/lexer.l:36-102.  Grammar and the order of the
 * semantic actions: reference src/parser.y:94-283.  See cs_frontend.h. */
#include "cs_frontend.h"

#include <ctype.h>
#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum tok {
  T_EOF = 0, T_NUM, T_IDENT, T_ANY, T_ALL, T_MIN, T_MAX, T_ALLDIFF,
  T_EQ, T_NEQ, T_LT, T_LEQ, T_GT, T_GEQ, T_MINUS, T_PLUS, T_STAR, T_BANG,
  T_AMP, T_BAR, T_LPAR, T_RPAR, T_COMMA, T_SEMI
};

typedef struct {
  const char *p;         /* cursor */
  unsigned line;         /* 1-based line of the current token */
  int tok;               /* look-ahead token */
  int32_t num;           /* value of T_NUM */
  char *ident;           /* text of T_IDENT (heap, reused) */
  size_t ident_cap;
  const cs_builder *b;
  char *err;
  size_t errlen;
  jmp_buf bail;
} parser;

static void fail(parser *ps, const char *fmt, const char *what) {
  if (ps->err != NULL && ps->errlen > 0) {
    char msg[160];
    snprintf(msg, sizeof msg, fmt, what);
    snprintf(ps->err, ps->errlen, "%s in line %u", msg, ps->line);
  }
  longjmp(ps->bail, 1);
}

static int is_sym_start(int c) { return c == '_' || c == '@' || c == '$' || isalpha(c); }
static int is_sym_char(int c) { return c == '_' || c == '@' || c == '$' || isalnum(c); }

/* longest-match scanning of one token, flex rule order resolved by hand */
static void advance(parser *ps) {
  const char *p = ps->p;
  for (;;) {
    while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') {
      if (*p == '\n') ps->line++;
      p++;
    }
    if (*p == '#') {
      while (*p != '\0' && *p != '\n') p++;
      continue;
    }
    break;
  }
  unsigned char c = (unsigned char)*p;
  if (c == '\0') { ps->tok = T_EOF; ps->p = p; return; }

  if (isdigit(c)) {
    const char *q = p;
    int base = 10;
    if (c == '0' && p[1] == 'b' && (p[2] == '0' || p[2] == '1')) {
      q = p + 2; base = 2;
      while (*q == '0' || *q == '1') q++;
      ps->num = (int32_t)strtol(p + 2, NULL, 2);
    } else if (c == '0' && (p[1] == 'x' || p[1] == 'X') && isxdigit((unsigned char)p[2])) {
      q = p + 2; base = 16;
      while (isxdigit((unsigned char)*q)) q++;
      ps->num = (int32_t)strtol(p + 2, NULL, 16);
    } else if (c == '0') {
      q = p + 1; base = 8;
      while (*q >= '0' && *q <= '7') q++;
      ps->num = (int32_t)strtol(p, NULL, 8);
    } else {
      while (isdigit((unsigned char)*q)) q++;
      ps->num = (int32_t)strtol(p, NULL, 10);
    }
    (void)base;
    ps->tok = T_NUM; ps->p = q; return;
  }

  if (is_sym_start(c)) {
    const char *q = p;
    while (is_sym_char((unsigned char)*q)) q++;
    size_t len = (size_t)(q - p);
    ps->p = q;
    if (len == 3 && memcmp(p, "ANY", 3) == 0) { ps->tok = T_ANY; return; }
    if (len == 3 && memcmp(p, "ALL", 3) == 0) { ps->tok = T_ALL; return; }
    if (len == 3 && memcmp(p, "MAX", 3) == 0) { ps->tok = T_MAX; return; }
    if (len == 3 && memcmp(p, "MIN", 3) == 0) { ps->tok = T_MIN; return; }
    if (len == 13 && memcmp(p, "all_different", 13) == 0) { ps->tok = T_ALLDIFF; return; }
    if (len + 1 > ps->ident_cap) {
      ps->ident_cap = (len + 1) * 2;
      ps->ident = (char *)realloc(ps->ident, ps->ident_cap);
      if (ps->ident == NULL) fail(ps, "%s", "out of memory");
    }
    memcpy(ps->ident, p, len);
    ps->ident[len] = '\0';
    ps->tok = T_IDENT; return;
  }

  ps->p = p + 1;
  switch (c) {
  case '=': ps->tok = T_EQ; return;
  case '!': if (p[1] == '=') { ps->p = p + 2; ps->tok = T_NEQ; } else ps->tok = T_BANG; return;
  case '<': if (p[1] == '=') { ps->p = p + 2; ps->tok = T_LEQ; } else ps->tok = T_LT; return;
  case '>': if (p[1] == '=') { ps->p = p + 2; ps->tok = T_GEQ; } else ps->tok = T_GT; return;
  case '-': ps->tok = T_MINUS; return;
  case '+': ps->tok = T_PLUS; return;
  case '*': ps->tok = T_STAR; return;
  case '&': ps->tok = T_AMP; return;
  case '|': ps->tok = T_BAR; return;
  case '(': ps->tok = T_LPAR; return;
  case ')': ps->tok = T_RPAR; return;
  case ',': ps->tok = T_COMMA; return;
  case ';': ps->tok = T_SEMI; return;
  default: {
    char bad[2] = { (char)c, '\0' };
    fail(ps, "invalid input `%s'", bad);
  }
  }
}

static void expect(parser *ps, int tok, const char *what) {
  if (ps->tok != tok) fail(ps, "syntax error, expecting %s", what);
  advance(ps);
}

static void *parse_expr(parser *ps);

/* PrimaryExpr (parser.y:135-152) */
static void *parse_primary(parser *ps) {
  const cs_builder *b = ps->b;
  if (ps->tok == T_NUM) {
    void *e = b->num(b->ctx, ps->num);
    advance(ps);
    return e;
  }
  if (ps->tok == T_IDENT) {
    void *e = b->ident(b->ctx, ps->ident);
    advance(ps);
    return e;
  }
  if (ps->tok == T_LPAR) {
    advance(ps);
    void *e = parse_expr(ps);
    expect(ps, T_RPAR, "')'");
    return e;
  }
  fail(ps, "syntax error, unexpected %s", "token");
  return NULL;
}

/* all_different(e1,...,en): the reference prepends to its list, so pairs are
 * enumerated over the reversed sequence (parser_support.c:275-284, parser.y:169-182) */
static void *parse_alldiff(parser *ps) {
  const cs_builder *b = ps->b;
  size_t n = 0, cap = 16;
  void **ex = (void **)malloc(cap * sizeof *ex);
  if (ex == NULL) fail(ps, "%s", "out of memory");
  expect(ps, T_LPAR, "'('");
  for (;;) {
    void *e = parse_expr(ps);
    if (n == cap) {
      cap *= 2;
      ex = (void **)realloc(ex, cap * sizeof *ex);
      if (ex == NULL) fail(ps, "%s", "out of memory");
    }
    ex[n++] = e;
    if (ps->tok != T_COMMA) break;
    advance(ps);
  }
  expect(ps, T_RPAR, "')'");

  size_t pairs = n * (n - 1) / 2, k = 0;
  void **el = (void **)malloc((pairs ? pairs : 1) * sizeof *el);
  if (el == NULL) fail(ps, "%s", "out of memory");
  for (size_t i = n; i-- > 0;) {
    for (size_t j = i; j-- > 0;) {
      void *eq = b->binary(b->ctx, CS_OP_EQ, ex[i], ex[j]);
      el[k++] = b->unary(b->ctx, CS_OP_NOT, eq);
    }
  }
  void *w = b->wand(b->ctx, el, pairs);
  free(el);
  free(ex);
  return w;
}

/* UnaryExpr (parser.y:154-185): '-' and '!' bind to a PrimaryExpr only */
static void *parse_unary(parser *ps) {
  const cs_builder *b = ps->b;
  if (ps->tok == T_MINUS) {
    advance(ps);
    return b->unary(b->ctx, CS_OP_NEG, parse_primary(ps));
  }
  if (ps->tok == T_BANG) {
    advance(ps);
    return b->unary(b->ctx, CS_OP_NOT, parse_primary(ps));
  }
  if (ps->tok == T_ALLDIFF) {
    advance(ps);
    return parse_alldiff(ps);
  }
  return parse_primary(ps);
}

/* MultExpr (194-199) */
static void *parse_mult(parser *ps) {
  const cs_builder *b = ps->b;
  void *l = parse_unary(ps);
  while (ps->tok == T_STAR) {
    advance(ps);
    void *r = parse_unary(ps);
    l = b->binary(b->ctx, CS_OP_MUL, l, r);
  }
  return l;
}

/* AddExpr (201-212): a-b is ADD(a, NEG(b)) */
static void *parse_add(parser *ps) {
  const cs_builder *b = ps->b;
  void *l = parse_mult(ps);
  while (ps->tok == T_PLUS || ps->tok == T_MINUS) {
    int minus = ps->tok == T_MINUS;
    advance(ps);
    void *r = parse_mult(ps);
    if (minus) r = b->unary(b->ctx, CS_OP_NEG, r);
    l = b->binary(b->ctx, CS_OP_ADD, l, r);
  }
  return l;
}

/* RelatExpr (214-243): only LT exists; <= >= > are rewritten */
static void *parse_relat(parser *ps) {
  const cs_builder *b = ps->b;
  void *l = parse_add(ps);
  for (;;) {
    int t = ps->tok;
    if (t != T_LT && t != T_GT && t != T_LEQ && t != T_GEQ) return l;
    advance(ps);
    void *r = parse_add(ps);
    void *e;
    switch (t) {
    case T_LT: e = b->binary(b->ctx, CS_OP_LT, l, r); break;
    case T_GT: e = b->binary(b->ctx, CS_OP_LT, r, l); break;
    case T_LEQ: e = b->unary(b->ctx, CS_OP_NOT, b->binary(b->ctx, CS_OP_LT, r, l)); break;
    default: e = b->unary(b->ctx, CS_OP_NOT, b->binary(b->ctx, CS_OP_LT, l, r)); break;
    }
    b->weigh(b->ctx, e, CS_WEIGHT_COMPARE);
    l = e;
  }
}

/* EqualExpr (245-265) */
static void *parse_equal(parser *ps) {
  const cs_builder *b = ps->b;
  void *l = parse_relat(ps);
  for (;;) {
    int t = ps->tok;
    if (t != T_EQ && t != T_NEQ) return l;
    advance(ps);
    void *r = parse_relat(ps);
    void *e = b->binary(b->ctx, CS_OP_EQ, l, r);
    if (t == T_NEQ) {
      e = b->unary(b->ctx, CS_OP_NOT, e);
      b->weigh(b->ctx, e, CS_WEIGHT_NOT_EQUAL);
    } else {
      b->weigh(b->ctx, e, CS_WEIGHT_EQUAL);
    }
    l = e;
  }
}

/* AndExpr (267-272) */
static void *parse_and(parser *ps) {
  const cs_builder *b = ps->b;
  void *l = parse_equal(ps);
  while (ps->tok == T_AMP) {
    advance(ps);
    void *r = parse_equal(ps);
    l = b->binary(b->ctx, CS_OP_AND, l, r);
  }
  return l;
}

/* OrExpr / Expr (274-283) */
static void *parse_expr(parser *ps) {
  const cs_builder *b = ps->b;
  void *l = parse_and(ps);
  while (ps->tok == T_BAR) {
    advance(ps);
    void *r = parse_and(ps);
    l = b->binary(b->ctx, CS_OP_OR, l, r);
  }
  return l;
}

int cs_parse_text(const char *text, const cs_builder *b, char *err, size_t errlen) {
  parser ps;
  memset(&ps, 0, sizeof ps);
  ps.p = text;
  ps.line = 1;
  ps.b = b;
  ps.err = err;
  ps.errlen = errlen;
  if (err != NULL && errlen > 0) err[0] = '\0';
  if (setjmp(ps.bail) != 0) {
    free(ps.ident);
    return -1;
  }
  advance(&ps);

  /* Objective (parser.y:109-131) opens the input */
  int kind;
  void *oexpr = NULL;
  switch (ps.tok) {
  case T_ANY: kind = CS_OBJ_ANY; advance(&ps); break;
  case T_ALL: kind = CS_OBJ_ALL; advance(&ps); break;
  case T_MIN: kind = CS_OBJ_MIN; advance(&ps); oexpr = parse_expr(&ps); break;
  case T_MAX: kind = CS_OBJ_MAX; advance(&ps); oexpr = parse_expr(&ps); break;
  default: fail(&ps, "syntax error, expecting %s", "ANY or ALL or MIN or MAX"); return -1;
  }
  expect(&ps, T_SEMI, "';'");
  b->constraint(b->ctx, b->objective(b->ctx, kind, oexpr));

  /* Constraints (94-106, 133) */
  while (ps.tok != T_EOF) {
    void *e = parse_expr(&ps);
    expect(&ps, T_SEMI, "';'");
    b->constraint(b->ctx, e);
  }
  free(ps.ident);
  return 0;
}
