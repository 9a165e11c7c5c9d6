/* cs_main.c -- command-line front for the GPU engine with the reference's input and output
 * conventions:
 *   input  : the csolve problem text (file argument, "-" or none = stdin), reference src/main.c:313-322
 *   output : "INFEASIBLE PROBLEM" (parser.y:70-72), one line per solution
 *            "#1: SOLUTION: X1 = 8, X2 = 2, ..., BEST: 0"  (print.c:24-31, 49-70, csolve.c:233-234),
 *            a final statistics line "#1: CALLS: n, CUTS: n, PROPS: n, CONFL: 0, RESTARTS: n, ...,
 *            SOLUTIONS: n" (csolve.c:54-58, csolve.h:469-479) and "NO SOLUTION FOUND" (csolve.c:184-186)
 *   errors : "<argv0>: error: <message>" on stderr, exit status 1 (print.c:73-94)
 * Options (reference src/main.c:51-130): -w <bool> weights; -o <order> none | smallest-domain | largest-domain |
 * smallest-value | largest-value (the engine's default is smallest-domain, the reference's none); -f <bool> prefer
 * failing variables; -r <int> restart frequency (ANY: Luby restarts every r x 64-parent iterations, 0 = none; MIN /
 * MAX: r > 0 also restarts on every better solution, csolve.c:418-425); -t <int> time limit in seconds (0 = none);
 * -c <bool> is accepted (this engine does not learn conflict clauses: the drop-in does); -j <int> is
 * accepted (one engine per process: ranks are started by the launcher, csolve_amd/parallel.py); -s -b -p -m -M size
 * the reference's host structures and are accepted and ignored.
 * The search order differs from the reference's (batched expansion), so CALLS/CUTS and WHICH
 * solution an ANY run prints are engine-specific; the set of solutions and the optimum are not.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/csolve_gpu.h"

static const char *prog = "csolve_gpu";

static void die(const char *msg) {
  fprintf(stderr, "%s: error: %s\n", prog, msg);
  exit(EXIT_FAILURE);
}

static char *read_all(FILE *f) {
  size_t cap = 1 << 16, n = 0;
  char *buf = (char *)malloc(cap);
  for (;;) {
    size_t got = fread(buf + n, 1, cap - n - 1, f);
    n += got;
    if (got == 0) break;
    if (n + 1 >= cap) buf = (char *)realloc(buf, cap *= 2);
  }
  buf[n] = '\0';
  return buf;
}

int main(int argc, char **argv) {
  prog = argv[0];
  int weights = 1, order = -1, prefer = 0, have_strategy = 0;
  long restart_freq = -1, time_max = 0;
  const char *path = NULL;
  for (int i = 1; i < argc; i++) {
    if (argv[i][0] == '-' && argv[i][1] != '\0') {
      if (strchr("bcfjmMoprstw", argv[i][1]) == NULL || i + 1 >= argc)
        die("usage: csolve_gpu [-w <bool>] [-o <order>] [-f <bool>] [-r <int>] [-t <seconds>] [-c <bool>] [-j <int>] [<file>]");
      const char *arg = argv[i + 1];
      switch (argv[i][1]) {
      case 'w': weights = strcmp(arg, "true") == 0; break;
      case 'f': prefer = strcmp(arg, "true") == 0; have_strategy = 1; break;
      case 'o':
        order = strcmp(arg, "none") == 0 ? 0 : strcmp(arg, "smallest-domain") == 0 ? 1 : strcmp(arg, "largest-domain") == 0 ? 2 :
                strcmp(arg, "smallest-value") == 0 ? 3 : strcmp(arg, "largest-value") == 0 ? 4 : -2;
        if (order == -2) die("invalid order"); /* ERROR_MSG_INVALID_STRATEGY_ORDER */
        have_strategy = 1;
        break;
      case 'r': restart_freq = strtol(arg, NULL, 10); break;
      case 't': time_max = strtol(arg, NULL, 10); break;
      default: break; /* -c -j -s -b -p -m -M: accepted */
      }
      i++;
    } else {
      path = argv[i];
    }
  }
  FILE *in = stdin;
  if (path != NULL && strcmp(path, "-") != 0 && (in = fopen(path, "r")) == NULL) die("cannot open input");
  char *text = read_all(in);

  csgpu_model *m = NULL;
  if (csgpu_model_from_text(text, weights, &m) != CSGPU_OK) die(csgpu_last_error());
  free(text);
  int32_t st = 0;
  if (csgpu_model_root_propagate(m, &st) != CSGPU_OK) die(csgpu_last_error());
  if (st >= 0) {
    if (csgpu_model_normalize(m) != CSGPU_OK || csgpu_model_root_propagate(m, &st) != CSGPU_OK) die(csgpu_last_error());
  }
  if (st < 0) {
    printf("INFEASIBLE PROBLEM\n");
    return EXIT_SUCCESS;
  }
  if (csgpu_model_finalize(m) != CSGPU_OK) die(csgpu_last_error());

  const int n = csgpu_model_num_vars(m);
  csgpu_search *s = NULL;
  if (csgpu_search_create(m, 1 << 21, 1 << 17, &s) != CSGPU_OK) die(csgpu_last_error());
  csgpu_val *root = (csgpu_val *)malloc((size_t)n * sizeof *root);
  csgpu_model_get_domains(m, root);
  if (have_strategy && csgpu_search_set_strategy(s, order < 0 ? 1 : order, prefer) != CSGPU_OK) die(csgpu_last_error());
  if (restart_freq >= 0) {
    if (csgpu_search_set_restart(s, restart_freq) != CSGPU_OK) die(csgpu_last_error());
    if (csgpu_search_set_restart_on_improvement(s, restart_freq > 0) != CSGPU_OK) die(csgpu_last_error());
  }
  if (csgpu_search_put_host(s, root, 1) != CSGPU_OK) die(csgpu_last_error());
  csgpu_search_stats stats;
  if (time_max <= 0) {
    if (csgpu_search_run(s, (int64_t)1 << 60, &stats) != CSGPU_OK) die(csgpu_last_error());
  } else {
    /* -t: the clock is looked at between slices of 64 iterations (the reference's SIGALRM sets a flag its loop tests) */
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
      if (csgpu_search_run(s, 64, &stats) != CSGPU_OK) die(csgpu_last_error());
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if (stats.done || (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) >= (double)time_max) break;
    }
  }

  const int64_t shown = (int64_t)(stats.solutions < 1024 ? stats.solutions : 1024);
  int32_t *vals = (int32_t *)malloc((size_t)(shown ? shown : 1) * (size_t)n * sizeof(int32_t));
  int64_t k = csgpu_search_solutions(s, vals, shown);
  const int obj = csgpu_model_objective(m), ov = csgpu_model_objective_var(m);
  if (obj >= 2) /* MIN/MAX: the solution that attains the optimum (the reference's last line) */
    k = csgpu_search_best_solution(s, vals) == 1 ? 1 : 0;
  for (int64_t i = 0; i < k; i++) {
    printf("#1: SOLUTION: ");
    for (int v = 0; v < n; v++) printf("%s = %d, ", csgpu_model_var_name(m, v), vals[i * n + v]);
    printf("BEST: %d\n", (obj >= 2 && ov >= 0) ? vals[i * n + ov] : 0);
  }
  printf("#1: CALLS: %lu, CUTS: %lu, PROPS: %lu, CONFL: 0, RESTARTS: %lu, LEVEL: 0/%d, AVG LEVEL: 0.000000, MEM: 0, CMEM: 0, SOLUTIONS: %lu\n",
         (unsigned long)stats.nodes, (unsigned long)stats.cuts, (unsigned long)stats.props,
         (unsigned long)stats.restarts, n, (unsigned long)stats.solutions);
  if (stats.solutions == 0) printf("NO SOLUTION FOUND\n");
  csgpu_search_free(s);
  csgpu_model_free(m);
  free(root); free(vals);
  return EXIT_SUCCESS;
}
