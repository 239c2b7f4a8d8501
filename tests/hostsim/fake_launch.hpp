// fake_launch.hpp -- switches of the launcher stand-ins (fake_launch.cpp; TEST INFRASTRUCTURE ONLY)
#pragma once

namespace fakelaunch {

// the scalars the host state machines branch on, as the stand-ins report them
struct Script {
	bool reject_step = false;      // the guard of the next guarded update says "non-finite / too long": x untouched, ring flushed
	bool reject_pair = false;      // with min_curvature > 0: the next pair fails the curvature test (rollback)
	double sy = 1.0, ss = 1.0, yy = 1.0;   // (s'y, s's, y'y) of a new pair: kappa = sqrt(ss) sqrt(yy) / |sy|
};
Script& script();

}  // namespace fakelaunch
