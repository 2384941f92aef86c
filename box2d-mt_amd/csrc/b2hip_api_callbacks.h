// b2hip_api_callbacks.h - part of the ONE translation unit b2hip.hip, inside its extern "C" block: listener / filter callbacks,
// island labels, fat AABBs, debug reads, profile, counters and the kernel-timing hooks bench.py uses.
// (No include guard on purpose: b2hip.hip includes it exactly once, in order - the fragments share one scope.)

int b2hip_set_contact_filter(b2hip_world* w, b2hip_should_collide_fn fn, void* user)
{
	if (int rcu = checkUsable(w, "b2hip_set_contact_filter", true)) return rcu;
	w->filterFn = fn;
	w->filterUser = user;
	w->dw.userFilter = hasFilter(w) ? 1 : 0;
	return B2HIP_OK;
}

int b2hip_set_contact_filter_batch(b2hip_world* w, b2hip_should_collide_batch_fn fn, void* user)
{
	if (int rcu = checkUsable(w, "b2hip_set_contact_filter_batch", true)) return rcu;
	w->filterBatchFn = fn;
	if (fn) w->filterUser = user;
	w->dw.userFilter = hasFilter(w) ? 1 : 0;
	return B2HIP_OK;
}

int b2hip_default_should_collide(b2hip_world* w, int fixture_a, int fixture_b)
{
	if (!w || fixture_a < 0 || fixture_b < 0 || fixture_a >= (int)w->fixtures.size() || fixture_b >= (int)w->fixtures.size())
		return setError(B2HIP_ERR_INVALID, "bad fixture id");
	// b2ContactFilter::ShouldCollide (b2WorldCallbacks.cpp:24-38)
	const HostFixture& a = w->fixtures[fixture_a];
	const HostFixture& b = w->fixtures[fixture_b];
	if (a.groupIndex == b.groupIndex && a.groupIndex != 0) return a.groupIndex > 0 ? 1 : 0;
	return ((a.maskBits & b.categoryBits) != 0 && (a.categoryBits & b.maskBits) != 0) ? 1 : 0;
}

int b2hip_set_pre_solve(b2hip_world* w, b2hip_pre_solve_fn fn, void* user)
{
	if (int rcu = checkUsable(w, "b2hip_set_pre_solve", true)) return rcu;
	w->preSolveFn = fn;
	w->preSolveUser = user;
	w->dw.preSolveOn = hasPreSolve(w) ? 1 : 0;
	return B2HIP_OK;
}

int b2hip_set_pre_solve_batch(b2hip_world* w, b2hip_pre_solve_batch_fn fn, void* user)
{
	if (int rcu = checkUsable(w, "b2hip_set_pre_solve_batch", true)) return rcu;
	w->preSolveBatchFn = fn;
	if (fn) w->preSolveUser = user;
	w->dw.preSolveOn = hasPreSolve(w) ? 1 : 0;
	return B2HIP_OK;
}

int b2hip_enable_post_solve(b2hip_world* w, int enable)
{
	if (int rcu = checkUsable(w, "b2hip_enable_post_solve", true)) return rcu;
	w->postSolveOn = enable != 0;
	w->dw.postSolveOn = enable ? 1 : 0;
	w->postSolve.clear();
	return B2HIP_OK;
}

int b2hip_get_post_solve(b2hip_world* w, int cap, b2hip_contact_impulse* out)
{
	if (!w || (cap > 0 && !out)) return setError(B2HIP_ERR_INVALID, "null argument");
	const int n = (int)w->postSolve.size();
	for (int i = 0; i < n && i < cap; ++i) out[i] = w->postSolve[i];
	return n;
}

int b2hip_get_island_labels(b2hip_world* w, int cap, int32_t* out)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	DEVICE_GUARD(w);
	if (!w || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	const int n = std::min(cap, (int)w->bodies.size());
	if (n <= 0) return 0;
	std::vector<int> parent(n), tier(n);
	HIP_TRY(hipMemcpy(parent.data(), w->parent.p, n * sizeof(int), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(tier.data(), w->rootIsland.p, n * sizeof(int), hipMemcpyDeviceToHost));
	for (int i = 0; i < n; ++i)
	{
		const HostBody& b = w->bodies[i];
		if (b.type == B2HIP_STATIC_BODY) { out[i] = -1; continue; }
		int r = parent[i];
		out[i] = (r >= 0 && r < n && tier[r] != ROOT_NONE) ? r : -1;
	}
	return n;
}

int b2hip_get_fat_aabb(b2hip_world* w, int fixture, float out4[4])
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	DEVICE_GUARD(w);
	if (!w || !out4 || fixture < 0 || fixture >= (int)w->fixtures.size()) return setError(B2HIP_ERR_INVALID, "bad fixture id");
	if ((size_t)fixture >= w->upFixtures)
	{
		memcpy(out4, w->fixtures[fixture].fat, 16);
		return 0;
	}
	HIP_TRY(hipMemcpy(out4, w->p_fat.p + fixture, 16, hipMemcpyDeviceToHost));
	return 0;
}

int b2hip_get_fat_aabbs(b2hip_world* w, int first, int count, float* out4n)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	DEVICE_GUARD(w);
	if (!w || (count > 0 && !out4n) || first < 0 || count < 0 || (size_t)(first + count) > w->fixtures.size())
		return setError(B2HIP_ERR_INVALID, "bad fixture range");
	const int onDevice = std::max(0, std::min(first + count, (int)w->upFixtures) - first);
	if (onDevice > 0) HIP_TRY(hipMemcpy(out4n, w->p_fat.p + first, (size_t)onDevice * 16, hipMemcpyDeviceToHost));
	for (int i = onDevice; i < count; ++i) memcpy(out4n + 4 * (size_t)i, w->fixtures[first + i].fat, 16); // not uploaded yet
	return 0;
}

// Debug / test hook: FNV-1a over a group of device arrays, read back after a stream sync. Valid between
// phase calls (b2hip_collide ... b2hip_step_end), so two worlds can be compared phase by phase.
//   which 0: bodies (pos, pos0, vel, xf, flags)   1: contacts (ids, key, flags & 0x7f, manifold, impulses)
//         2: proxies (fat AABBs)                  3: contact impulses only
static uint64_t fnv(uint64_t h, const void* data, size_t n)
{
	const unsigned char* p = (const unsigned char*)data;
	for (size_t i = 0; i < n; ++i)
	{
		h ^= p[i];
		h *= 1099511628211ull;
	}
	return h;
}

int b2hip_debug_hash(b2hip_world* w, int which, uint64_t* out)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	DEVICE_GUARD(w);
	if (!w || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	HIP_TRY(hipStreamSynchronize(w->stream));
	DState st;
	HIP_TRY(hipMemcpy(&st, w->d_state.p, sizeof(DState), hipMemcpyDeviceToHost));
	uint64_t h = 1469598103934665603ull;
	std::vector<unsigned char> buf;
	auto pull = [&](const void* dev, size_t bytes) -> int
	{
		buf.resize(bytes);
		if (bytes == 0) return 0;
		if (hipMemcpy(buf.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
		h = fnv(h, buf.data(), bytes);
		return 0;
	};
	const size_t nb = w->upBodies, np = w->upFixtures, nc = (size_t)st.c.nContacts;
	const int cur = st.cur;
	int bad = 0;
	if (which == 0)
	{
		bad |= pull(w->b_pos.p, nb * 16); bad |= pull(w->b_pos0.p, nb * 16); bad |= pull(w->b_vel.p, nb * 16);
		bad |= pull(w->b_xf.p, nb * 16);
		std::vector<uint32_t> f(nb);
		if (nb && hipMemcpy(f.data(), w->b_flags.p, nb * 4, hipMemcpyDeviceToHost) != hipSuccess) bad = 1;
		for (size_t i = 0; i < nb; ++i) f[i] &= 0x7fu;
		h = fnv(h, f.data(), nb * 4);
	}
	else if (which == 1)
	{
		bad |= pull(w->c_ids[cur].p, nc * 16); bad |= pull(w->c_key[cur].p, nc * 8);
		std::vector<uint32_t> f(nc);
		if (nc && hipMemcpy(f.data(), w->c_flags[cur].p, nc * 4, hipMemcpyDeviceToHost) != hipSuccess) bad = 1;
		for (size_t i = 0; i < nc; ++i) f[i] &= 0x1fu;
		h = fnv(h, f.data(), nc * 4);
		bad |= pull(w->c_man0[cur].p, nc * 16); bad |= pull(w->c_man1[cur].p, nc * 16);
		bad |= pull(w->c_imp[cur].p, nc * 16); bad |= pull(w->c_man3[cur].p, nc * 16);
	}
	else if (which == 2)
	{
		bad |= pull(w->p_fat.p, np * 16);
	}
	else
	{
		bad |= pull(w->c_imp[cur].p, nc * 16);
	}
	if (bad) return setError(B2HIP_ERR_HIP, "debug hash read-back failed");
	*out = h;
	return 0;
}

// Debug hook: raw read of a device array (0 b_pos, 1 b_vel, 2 li_bodies, 3 b_force, 4 b_flags, 5 b_damp, 6 b_mass,
// 7 counters as ints) into `out` (bytes).
int b2hip_debug_read(b2hip_world* w, int which, int first, int count, void* out)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	DEVICE_GUARD(w);
	if (!w || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	HIP_TRY(hipStreamSynchronize(w->stream));
	const void* src = nullptr;
	size_t elem = 16;
	switch (which)
	{
	case 0: src = w->b_pos.p; break;
	case 1: src = w->b_vel.p; break;
	case 2: src = w->li_bodies.p; elem = 4; break;
	case 3: src = w->b_force.p; break;
	case 4: src = w->b_flags.p; elem = 4; break;
	case 5: src = w->b_damp.p; break;
	case 6: src = w->b_mass.p; break;
	case 7: src = w->d_state.p; elem = 4; break;
	case 8: src = w->dbgPreVel.p; break;
	case 9: src = w->dbgVel.p; break;
	case 10: src = w->dbgLi.p; elem = 4; break;
	case 11: src = w->gridBar.p; elem = 4; break;
	case 12:
	{
		// (the colour census: counter c at colorSlot(c) - the first 65 a 128-byte line apart)
		if (first < 0 || count < 0 || first + count > COLOR_SLOT_PADDED) return setError(B2HIP_ERR_INVALID, "colour census: [0, 65)");
		HIP_TRY(hipStreamSynchronize(w->stream));
		if (count > 0) HIP_TRY(hipMemcpy2D(out, sizeof(int), w->colorCount.p + colorSlot(first), COLOR_SLOT_STRIDE * sizeof(int), sizeof(int), (size_t)count, hipMemcpyDeviceToHost));
		return 0;
	}
	case 13: src = w->bodyColorMask.p; elem = 8; break;
	case 14: src = w->deg.p; elem = 4; break;
	case 15: src = w->hubList.p; elem = 4; break;
	case 16: src = w->bodyActive.p; elem = 8; break;
	case 17: src = w->b_blk1.p; elem = 4; break;
	case 18: src = w->li_ref.p; break;
	case 19: src = w->rowColor.p; elem = 4; break;
	case 20: src = w->blkRowStart.p; elem = 4; break;
	case 21: src = w->bodyRest.p; elem = 8; break;
	default: return setError(B2HIP_ERR_INVALID, "bad array id");
	}
	HIP_TRY(hipMemcpy(out, (const char*)src + (size_t)first * elem, (size_t)count * elem, hipMemcpyDeviceToHost));
	return 0;
}

// Debug hook (B2HIP_TRACE=1): stage labels + state hashes recorded by the last b2hip_solve.
int b2hip_debug_trace(b2hip_world* w, int index, char* label, int label_cap, uint64_t* hash)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	if (index < 0 || index >= (int)w->trace.size()) return 1;
	if (label && label_cap > 0)
	{
		strncpy(label, w->trace[index].first.c_str(), (size_t)label_cap - 1);
		label[label_cap - 1] = 0;
	}
	if (hash) *hash = w->trace[index].second;
	return 0;
}

int b2hip_get_profile(b2hip_world* w, float ms[13])
{
	if (!w || !ms) return setError(B2HIP_ERR_INVALID, "null argument");
	memcpy(ms, w->profile, sizeof(float) * 13);
	return 0;
}

int b2hip_get_counters(b2hip_world* w, b2hip_counters* out)
{
	if (!w || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	memset(out, 0, sizeof(*out));
	out->bodies = (int)w->bodies.size();
	out->proxies = (int)w->fixtures.size();
	out->contacts = w->last.nContacts;
	out->touching_contacts = w->last.nTouching;
	out->islands = w->last.nIslands;
	out->small_islands = w->last.nSIslands;
	out->large_islands = w->last.nLIslands;
	out->small_island_bodies = w->last.nSBodies;
	out->small_island_contacts = w->last.nSContacts;
	out->large_island_bodies = w->last.nLBodies;
	out->large_island_contacts = w->last.nLContacts;
	out->colors = w->last.nColors;
	out->moved_proxies = w->last.nMoves;
	out->new_contacts = w->last.nNewContacts;
	out->destroyed_contacts = w->last.nDestroy;
	out->solver_chunks = w->last.nChunks;
	out->pos_iterations_large = w->last.posItersLarge;
	out->overflow_flags = w->last.overflow;
	out->toi_events = w->last.nToiEvents;
	out->toi_calls = w->last.nToiCalls;
	out->toi_pending_first_pass = w->last.nToiList;
	out->toi_serial_fallbacks = w->toiFallbacks;
	out->blocks = w->last.nBlocks;
	out->cut_constraints = w->last.nCutRows;
	out->block_max_rows = w->last.blkMaxRows;
	out->partitions = w->last.partitions;
	out->block_solver_steps = w->blockSteps;
	out->free_islands = w->last.nFreeIslands;
	out->sweep_solver_steps = w->sweepSteps;
	out->hub_constraints = w->last.nHubRows;
	out->hub_fixpoint_rounds = w->last.hubRounds;
	out->hub_serial_chunks = w->last.hubSerialChunks;
	out->toi_chain_contacts = w->toiChainContacts;
	out->toi_pre_solve_reruns = w->toiPreSolveReruns;
	out->solver_recoveries = w->solverRecoveries + w->colorRecoveries;
	return 0;
}

int b2hip_get_solver_timing(b2hip_world* w, float* ms, double* algorithmic_bytes, int* constraints, int* bodies)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	if (ms) *ms = w->solverMs;
	if (algorithmic_bytes) *algorithmic_bytes = w->solverBytes;
	if (constraints) *constraints = w->solverConstraints;
	if (bodies) *bodies = w->solverBodies;
	return 0;
}

int b2hip_set_kernel_timing(b2hip_world* w, int enable)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	w->kernelTiming = enable;
	return 0;
}

int b2hip_set_kernel_timing_units(b2hip_world* w, long long units_a, long long units_b)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	w->ktUnitsA = units_a;
	w->ktUnitsB = units_b;
	return 0;
}

int b2hip_get_kernel_timing(b2hip_world* w, char* name, int name_cap, float* total_ms, int* launches, double* algorithmic_bytes)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	const char* n = w->ktKind == 8 ? "large-island solver family (k_large_integrate / init / velocity / rest / k_sweep_end / position / store_impulses / finalize / sleep)" : w->ktKind == 5 ? "k_collide" : w->ktKind == 6 ? "k_sync_fixtures" : w->ktKind == 7 ? "k_find_pairs_small" : w->ktKind == 4 ? "k_solve_blocks" : w->ktKind == 1 ? "k_large_velocity" : (w->ktKind == 2 ? "k_solve_small" : (w->ktKind == 3 ? (w->solverBarriers ? "k_solve_persistent" : (w->solverRows ? "k_solve_dataflow" : "k_solve_mailbox")) : ""));
	if (name && name_cap > 0)
	{
		strncpy(name, n, (size_t)name_cap - 1);
		name[name_cap - 1] = 0;
	}
	if (total_ms) *total_ms = w->ktMs;
	if (launches) *launches = w->ktLaunches;
	if (algorithmic_bytes) *algorithmic_bytes = w->ktBytes;
	return 0;
}

// (diagnostics, not part of include/b2hip.h: the device clock stamps around the mid-step census read-back, in 10 ns ticks)
// (diagnostics: the six phase stamps of the last step's resident solver - or, with B2HIP_SWEEP_STAMPS=1, of its last k_sweep_end<1>)
int b2hip_debug_stamps(b2hip_world* w, int out[6])
{
	if (!w || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	for (int k = 0; k < 6; ++k) out[k] = w->h_dstate->stamps[k];
	return 0;
}

int b2hip_debug_gap_clocks(b2hip_world* w, unsigned long long out[4])
{
	if (!w || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	for (int k = 0; k < 4; ++k) out[k] = w->h_dstate->gapClock[k];
	return 0;
}

