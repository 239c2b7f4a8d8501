// The header-only C++ front-ends of include/stochqn.h (class names, constructor defaults, run(),
// get_*() as in reference include/stochqn.h:397-511) driven from ordinary C++ with std::vector
// storage.  Prints the final iterates with 17 digits for tests/test_c_callers.py.
//
//   g++ -std=c++11 -I include tests/c/raii_callers.cpp -L stochqn_amd/lib -lstochqn
#include <cstdio>
#include <vector>
#include "stochqn.h"

static const int N = 9;

static void gradient(const double* x, std::vector<double>& g, int call)
{
	for (int i = 0; i < N; i++) g[i] = (0.5 + 0.25 * i) * x[i] * (1.0 + 0.01 * (((call * 7 + i * 3) % 11) - 5) / 5.0);
}

static double objective(const double* x)
{
	double f = 0;
	for (int i = 0; i < N; i++) f += 0.5 * (0.5 + 0.25 * i) * x[i] * x[i];
	return f;
}

template <class Opt> static void report(const char* name, Opt& o, const std::vector<double>& x)
{
	std::printf("%s niter %zu task %d info %d", name, o.get_n_iter(), (int) o.get_task(), (int) o.get_iter_info());
	for (int i = 0; i < N; i++) std::printf(" %.17g", x[i]);
	std::printf("\n");
}

int main()
{
	{
		oLBFGS opt(N, 5);                                    // defaults: hess_init 0, y_reg 0, min_curvature 0, check_nan 1
		std::vector<double> x(N, 1.5), g(N);
		int last = 0;
		for (int call = 0; call < 81; call++) {
			opt.run(0.1, x.data(), g.data());
			if (opt.get_task() == calc_grad) last = call;
			gradient(opt.get_req(), g, opt.get_task() == calc_grad_same_batch ? last : call);
		}
		report("oLBFGS", opt, x);
	}
	{
		SQN opt(N, 4, 5, 1e-4, 1);                           // gradient-difference pairs
		std::vector<double> x(N, 1.5), g(N), hv(1);
		for (int call = 0; call < 90; call++) {
			opt.run(0.1, x.data(), g.data(), hv.data());
			gradient(opt.get_req(), g, call);
		}
		report("SQN", opt, x);
	}
	{
		adaQN opt(N, 4, 6, 5);                               // defaults: max_incr 1.01, rmsprop 0.9, Fisher pairs
		std::vector<double> x(N, 1.5), g(N);
		double f = 0;
		for (int call = 0; call < 90; call++) {
			opt.run(0.05, x.data(), f, g.data());
			if (opt.get_task() == calc_fun_val_batch) f = objective(opt.get_req());
			else gradient(opt.get_req(), g, call);
		}
		report("adaQN", opt, x);
	}
	return 0;
}
