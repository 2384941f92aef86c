// b2hip_api_step.h - part of the ONE translation unit b2hip.hip, inside its extern "C" block: b2hip_step and its phase entry
// points, the end of a step (fallbacks of the parallel TOI paths, profile, counters), state / contact event getters.
// (No include guard on purpose: b2hip.hip includes it exactly once, in order - the fragments share one scope.)

int b2hip_step_begin(b2hip_world* w, float dt, int velocity_iterations, int position_iterations)
{
	if (int rc = checkUsable(w, "b2hip_step_begin", false)) return rc;
	if (w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_step_begin inside a step (finish it with b2hip_step_end)");
	DEVICE_GUARD(w);
	return stepFailed(w, stepBeginImpl(w, dt, velocity_iterations, position_iterations));
}

static bool keyLess(const std::pair<unsigned long long, int>& a, const std::pair<unsigned long long, int>& b) { return a < b; }

static void toManifold(b2hip_manifold* m, float4 m0, float4 m1, float4 imp, int4 m3)
{
	m->type = m3.z;
	m->point_count = m3.w;
	m->local_normal[0] = m0.x; m->local_normal[1] = m0.y;
	m->local_point[0] = m0.z; m->local_point[1] = m0.w;
	m->point_local[0][0] = m1.x; m->point_local[0][1] = m1.y;
	m->point_local[1][0] = m1.z; m->point_local[1][1] = m1.w;
	m->normal_impulse[0] = imp.x; m->tangent_impulse[0] = imp.y;
	m->normal_impulse[1] = imp.z; m->tangent_impulse[1] = imp.w;
	m->id_key[0] = (uint32_t)m3.x;
	m->id_key[1] = (uint32_t)m3.y;
}

static int collideImpl(b2hip_world* w)
{
	int rc = 0;
	if (hasFilter(w) && w->refilterPending)
	{
		// b2ContactManager::Collide's re-filter (:195-203) with a user filter: the flagged contacts are shown to it first
		LAUNCH(w, k_filter_list, gridFor(w->dw.capContacts), 256, w->dw);
		rc = readState(w);
		if (rc) return rc;
		const int n = std::min(w->h_dstate->c.nFilterList, w->dw.capContacts);
		if (n > 0)
		{
			std::vector<int> list(n), refused;
			HIP_TRY(hipMemcpy(list.data(), w->filterList.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
			std::sort(list.begin(), list.end());
			const int cur = w->h_dstate->cur;
			std::vector<int> asked, verdict;
			for (int k = 0; k < n; ++k)
			{
				int4 ids;
				HIP_TRY(hipMemcpy(&ids, w->c_ids[cur].p + list[k], sizeof(int4), hipMemcpyDeviceToHost));
				asked.push_back(ids.x);
				asked.push_back(ids.y);
			}
			askFilter(w, asked, verdict);
			for (int k = 0; k < n; ++k) if (!verdict[k]) refused.push_back(list[k]);
			rc = applyHostList(w, k_filter_reject, refused);
			if (rc) return rc;
		}
	}
	w->refilterPending = false;
	rc = phaseCollide(w);
	if (rc) return rc;
	if (hasPreSolve(w))
	{
		// b2ContactListener::PreSolve: one record per touching, non-sensor contact this Collide updated; delivered in
		// proxy-id-pair order (b2ContactManager.cpp:431-434); a zero return disables the contact for this step
		rc = readState(w);
		if (rc) return rc;
		const int n = std::min(w->h_dstate->c.nPreSolve, w->dw.capContacts);
		if (n > 0)
		{
			std::vector<PreSolveRec> recs(n);
			HIP_TRY(hipMemcpy((void*)recs.data(), w->preRecs.p, (size_t)n * sizeof(PreSolveRec), hipMemcpyDeviceToHost));
			std::vector<std::pair<unsigned long long, int> > order(n);
			for (int i = 0; i < n; ++i) order[i] = std::make_pair(recs[i].key, i);
			std::sort(order.begin(), order.end(), keyLess);
			std::vector<int> disabled, materials;
			w->callbackWindow = true;
			std::vector<b2hip_pre_solve_record> batch((size_t)n);
			for (int k = 0; k < n; ++k)
			{
				const PreSolveRec& r = recs[order[k].second];
				b2hip_pre_solve_record& b = batch[k];
				b.contact_index = r.info.x;
				b.fixture_a = r.info.y;
				b.fixture_b = r.info.z;
				b.enabled = 1;
				toManifold(&b.old_manifold, r.o0, r.o1, r.oimp, r.o3);
				toManifold(&b.manifold, r.n0, r.n1, r.nimp, r.n3);
				b.material.friction = r.mat.x;
				b.material.restitution = r.mat.y;
				b.material.tangent_speed = r.mat.z;
			}
			if (w->preSolveBatchFn) w->preSolveBatchFn(w->preSolveUser, n, batch.data());
			else for (int k = 0; k < n; ++k)
			{
				b2hip_pre_solve_record& b = batch[k];
				b.enabled = w->preSolveFn(w->preSolveUser, b.contact_index, b.fixture_a, b.fixture_b, &b.old_manifold, &b.manifold, &b.material) ? 1 : 0;
			}
			for (int k = 0; k < n; ++k)
			{
				const PreSolveRec& r = recs[order[k].second];
				const b2hip_contact_material& mat = batch[k].material;
				if (!batch[k].enabled) disabled.push_back(r.info.x);
				if (memcmp(&mat.friction, &r.mat.x, 4) != 0 || memcmp(&mat.restitution, &r.mat.y, 4) != 0 || memcmp(&mat.tangent_speed, &r.mat.z, 4) != 0)
				{
					int bits[3];
					memcpy(bits, &mat, sizeof(bits));
					materials.push_back(r.info.x);
					materials.insert(materials.end(), bits, bits + 3);
				}
			}
			w->callbackWindow = false;
			rc = applyHostList(w, k_presolve_disable, disabled);
			if (rc) return rc;
			if (!materials.empty())
			{
				HIP_TRY(hipMemcpyAsync(w->hostList.p, materials.data(), materials.size() * sizeof(int), hipMemcpyHostToDevice, w->stream));
				LAUNCH(w, k_presolve_material, gridFor(materials.size() / 4), 256, w->dw, (const int*)w->hostList.p, (int)(materials.size() / 4));
				HIP_TRY(hipStreamSynchronize(w->stream));
			}
		}
		// Edits made from inside PreSolve take effect at once, as in the reference, whose deferred callbacks run between
		// Collide and Solve (b2ContactManager::FinishCollide, b2ContactManager.cpp:387-441; Testbed/Tests/TunnelingTest.h
		// switches sensors, thick shapes and bullets there and expects this step's solvers to see it)
		if (!w->dirtyList.empty() || !w->editOps.empty() || !w->proxyEdits.empty() || !w->pendingMoves.empty())
		{
			rc = flushEdits(w);
			if (rc) return rc;
			rc = applyEditOps(w, false);
			if (rc) return rc;
		}
	}
	// the begin / end events of THIS phase are listed now (k_contact_events compares touching with what the host was told and
	// flips CF_REPORTED): what the TOI sub-steps change later in the step is logged by the sub-steps themselves, in order
	if (w->eventsOn && w->def.continuous && w->sp.dt > 0.0f) LAUNCH(w, k_contact_events, gridFor(w->dw.capContacts), 256, w->dw);
	stampPhase(w, 2);
	return 0;
}

int b2hip_collide(b2hip_world* w)
{
	if (int rc = checkUsable(w, "b2hip_collide", false)) return rc;
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_collide outside a step");
	DEVICE_GUARD(w);
	return stepFailed(w, collideImpl(w));
}

static int shardExchangeOnStream(b2hip_world* w);

static int solveImpl(b2hip_world* w)
{
	if (w->sp.dt > 0.0f && w->stepSolves) // (b2World.cpp:1668: m_stepComplete && step.dt > 0)
	{
		int rc = phaseSolve(w);
		if (rc) return rc;
		// a connected sharded world (b2hip_shard_connect): the islands the other ranks solved arrive here, on the stream
		if (w->shardComm != nullptr && !w->spatial && (w->dw.shardCount > 1 || w->shardLoopback))
		{
			rc = shardExchangeOnStream(w);
			if (rc) return rc;
		}
	}
	else
	{
		// (a call that continues an open step: no island build to apply the wake-ups Collide asked for)
		if (w->sp.dt > 0.0f) LAUNCH(w, k_wake_apply, gridFor(w->dw.nBodies), 256, w->dw);
		for (int k = 4; k <= 8; ++k) stampPhase(w, k);
	}
	if (w->postSolveOn && w->sp.dt > 0.0f && w->stepSolves) LAUNCH(w, k_postsolve_gather, gridFor(w->dw.capContacts), 256, w->dw);
	stampPhase(w, 3);
	return 0;
}

int b2hip_solve(b2hip_world* w)
{
	if (int rc = checkUsable(w, "b2hip_solve", false)) return rc;
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_solve outside a step");
	DEVICE_GUARD(w);
	return stepFailed(w, solveImpl(w));
}

static int syncFixturesImpl(b2hip_world* w)
{
	if (w->sp.dt > 0.0f && w->stepSolves)
	{
		int rc = phaseSyncFixtures(w);
		if (rc) return rc;
		// E1: what the other ranks' bodies did in Solve, and the fat AABBs their SynchronizeFixtures moved
		if (w->spatial) { rc = spExchangeState(w, 0); if (rc) return rc; }
		rc = forkEarlyRows(w);
		if (rc) return rc;
	}
	stampPhase(w, 9);
	return 0;
}

int b2hip_sync_fixtures(b2hip_world* w)
{
	if (int rc = checkUsable(w, "b2hip_sync_fixtures", false)) return rc;
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_sync_fixtures outside a step");
	DEVICE_GUARD(w);
	return stepFailed(w, syncFixturesImpl(w));
}

static int findNewContactsImpl(b2hip_world* w)
{
	if (w->sp.dt > 0.0f && w->stepSolves)
	{
		int rc = findNewContactsGraph(w);
		if (rc) return rc;
		rc = startEarlyRows(w); // (if the pair update has not sent them off itself)
		if (rc) return rc;
	}
	stampPhase(w, 10);
	return 0;
}

int b2hip_find_new_contacts(b2hip_world* w)
{
	if (int rc = checkUsable(w, "b2hip_find_new_contacts", false)) return rc;
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_find_new_contacts outside a step");
	DEVICE_GUARD(w);
	return stepFailed(w, findNewContactsImpl(w));
}

static int solveToiImpl(b2hip_world* w)
{
	w->toiRan = false;
	w->toiChains = false;
	w->toiSpeculative = false;
	w->toiSnapshotTaken = false;
	w->toiVerdicts.clear();
	w->dw.nToiVerdict = 0;
	w->last.nToiList = w->last.nToiCalls = w->last.nToiEvents = 0;
	if (w->def.continuous && w->sp.dt > 0.0f)
	{
		int rc = phaseToi(w);
		if (rc) return rc;
		// E4: the other ranks' TOI events (after this rank's phase has settled: fallbacks run here, not at the step's end)
		for (int attempt = 0; w->spatial; ++attempt)
		{
			rc = spAfterToi(w);
			if (rc <= 0) { if (rc) return rc; break; }
			// (1: an event reached over an ownership boundary - the phase was taken back and the owners merged: once more)
			if (attempt == 8) return setError(B2HIP_ERR_INVALID, "the TOI phase of a spatially sharded world keeps reaching over ownership boundaries");
			w->toiRan = false; w->toiChains = false; w->toiSpeculative = false; w->toiSnapshotTaken = false; w->toiCountersFresh = false;
			rc = phaseToi(w);
			if (rc) return rc;
		}
	}
	stampPhase(w, 12);
	w->toiEventValid = true;
	return 0;
}

int b2hip_solve_toi(b2hip_world* w)
{
	if (int rc = checkUsable(w, "b2hip_solve_toi", false)) return rc;
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_solve_toi outside a step");
	DEVICE_GUARD(w);
	return stepFailed(w, solveToiImpl(w));
}

static int uploadToiVerdicts(b2hip_world* w);

// B2HIP_HANDOVER_WHY=1: what the lane that gave up first was waiting for (b2d_handover.h: dataflowRun), and every constraint
// row of the large islands that touches either of its two bodies - colour, the block segment it was placed in, the bodies'
// home blocks and cut-colour masks - to stderr. Diagnostics for the block solvers' hand-over protocol.
static void handoverPostMortem(b2hip_world* w)
{
	(void)hipStreamSynchronize(w->stream);
	int bar[32];
	if (hipMemcpy(bar, w->gridBar.p, sizeof(bar), hipMemcpyDeviceToHost) != hipSuccess || bar[24] == 0) { fprintf(stderr, "[b2hip] hand-over post mortem: no record\n"); return; }
	const Counters& c = w->h_dstate->c;
	auto bodyOf = [&](int unit) -> int
	{
		if (unit == -1) return -1;
		const float4* bases[2] = { w->b_cutv.p, w->b_posv.p };
		for (int k = 0; k < 2; ++k)
		{
			const uint32_t b0 = (uint32_t)((uintptr_t)bases[k] >> 4);
			const uint32_t d = (uint32_t)unit - b0;
			if (d < (uint32_t)w->bodies.size()) return (int)d;
		}
		return -2;
	};
	{
		DState ds;
		if (hipMemcpy(&ds, w->d_state.p, sizeof(DState), hipMemcpyDeviceToHost) == hipSuccess && ds.dbgCensus[0] != 0)
			fprintf(stderr, "[b2hip]   k_block_census left body %d without a block: effBlk %d, b_blk1 %d, offer 0x%x, hash 0x%x, blocks %d, hash mod blocks %d, effBlk again %d\n", ds.dbgCensus[0] - 1, ds.dbgCensus[1], ds.dbgCensus[2], ds.dbgCensus[3], ds.dbgCensus[4], ds.dbgCensus[5], ds.dbgCensus[6], ds.dbgCensus[7]);
	}
	const int bodyA = bodyOf(bar[29]), bodyB = bodyOf(bar[30]);
	fprintf(stderr, "[b2hip] hand-over post mortem: workgroup %d lane %d waited for body %d at version 0x%x (saw 0x%x) and body %d at 0x%x (saw 0x%x); %d blocks of %d lanes, %d large-island constraints, %d colours\n",
		bar[31] / 4096, bar[31] % 4096, bodyA, bar[25], bar[26], bodyB, bar[27], bar[28], c.nBlocks, c.blkLanes, c.nLContacts, c.nColors);
	const int n = c.nLContacts;
	if (n <= 0) return;
	std::vector<int4> ref((size_t)n);
	std::vector<int> col((size_t)n), rowStart((size_t)MAX_BLOCKS + 2), blk1(w->bodies.size()), adopt(w->bodies.size());
	std::vector<unsigned long long> act(w->bodies.size());
	(void)hipMemcpy(ref.data(), w->li_ref.p, (size_t)n * sizeof(int4), hipMemcpyDeviceToHost);
	(void)hipMemcpy(col.data(), w->rowColor.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
	(void)hipMemcpy(rowStart.data(), w->blkRowStart.p, rowStart.size() * sizeof(int), hipMemcpyDeviceToHost);
	(void)hipMemcpy(blk1.data(), w->b_blk1.p, blk1.size() * sizeof(int), hipMemcpyDeviceToHost);
	(void)hipMemcpy(adopt.data(), w->b_adopt.p, adopt.size() * sizeof(int), hipMemcpyDeviceToHost);
	(void)hipMemcpy(act.data(), w->bodyActive.p, act.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
	std::vector<uint32_t> flags(w->bodies.size());
	std::vector<int> deg(w->bodies.size());
	(void)hipMemcpy(flags.data(), w->b_flags.p, flags.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
	(void)hipMemcpy(deg.data(), w->deg.p, deg.size() * sizeof(int), hipMemcpyDeviceToHost);
	std::vector<int> lib((size_t)std::max(c.nLBodies, 1));
	(void)hipMemcpy(lib.data(), w->li_bodies.p, lib.size() * sizeof(int), hipMemcpyDeviceToHost);
	{
		int zero = 0, first = -1;
		for (int k = 0; k < c.nLBodies; ++k) if (blk1[(size_t)lib[(size_t)k]] == 0) { if (first < 0) first = lib[(size_t)k]; ++zero; }
		std::vector<int> bstart((size_t)MAX_BLOCKS + 2);
		(void)hipMemcpy(bstart.data(), w->blkBodyStart.p, bstart.size() * sizeof(int), hipMemcpyDeviceToHost);
		fprintf(stderr, "[b2hip]   %d of the %d large-island bodies have no home block (the first: %d); home bodies listed by the census: %d; bodies in the world %zu, joints %d, busiest body %d contacts\n",
			zero, c.nLBodies, first, bstart[(size_t)std::min(c.nBlocks, MAX_BLOCKS)], w->bodies.size(), w->dw.nJoints, c.maxDegree);
	}
	auto describe = [&](int body)
	{
		int at = -1;
		for (int k = 0; k < c.nLBodies; ++k) if (lib[(size_t)k] == body) at = k;
		const int nbk = c.nBlocks < MAX_BLOCKS ? c.nBlocks : MAX_BLOCKS;
		fprintf(stderr, "[b2hip]       (place in the list of large-island bodies: %d of %d; a block by its own id would be %d)\n", at, c.nLBodies, nbk > 0 ? ownIdBlock(body, nbk) : 0);
		fprintf(stderr, "[b2hip]       body %d: flags 0x%x (type %u, large %d, awake %d), degree %d, home block %d, offer 0x%x, cut-colour mask 0x%llx\n", body, flags[(size_t)body], flags[(size_t)body] & BF_TYPE_MASK,
			(flags[(size_t)body] & BF_LARGE) ? 1 : 0, (flags[(size_t)body] & BF_AWAKE) ? 1 : 0, deg[(size_t)body], blk1[(size_t)body], adopt[(size_t)body], act[(size_t)body]);
	};
	for (int which = 0; which < 2; ++which)
	{
		const int body = which ? bodyB : bodyA;
		if (body < 0) continue;
		fprintf(stderr, "[b2hip]   body %d: home block %d, offer 0x%x, cut-colour mask 0x%llx (%d cut constraints expected)\n", body, blk1[(size_t)body], adopt[(size_t)body], act[(size_t)body], __builtin_popcountll(act[(size_t)body]));
		for (int row = 0; row < n; ++row)
		{
			const int4 q = ref[(size_t)row];
			const int a = q.y >= 0 ? q.y : -(q.y + 1), b = q.z >= 0 ? q.z : -(q.z + 1);
			if ((q.y >= 0 && a == body) || (q.z >= 0 && b == body))
			{
				int seg = -1;
				for (int k = 0; k <= c.nBlocks && k <= MAX_BLOCKS; ++k) if (row >= rowStart[(size_t)k] && row < rowStart[(size_t)k + 1]) seg = k;
				fprintf(stderr, "[b2hip]     row %d colour %d in the segment of block %d: contact %d, bodies %d%s (block %d) / %d%s (block %d)\n", row, col[(size_t)row], seg + 1, q.x,
					a, q.y >= 0 ? "" : " static", q.y >= 0 ? blk1[(size_t)a] : 0, b, q.z >= 0 ? "" : " static", q.z >= 0 ? blk1[(size_t)b] : 0);
				if (q.y >= 0) describe(a);
				if (q.z >= 0) describe(b);
			}
		}
	}
}

// A contact created inside a TOI sub-step did not fit the array (whatever path ran last, fallbacks included): never a
// silent drop. With the snapshot of this step's TOI phase at hand the phase is undone, the array doubled and the phase
// run again; without one (serial-only mode) it is an error.
static int settleToiOverflow(b2hip_world* w)
{
	int rc = 0;
	for (int attempt = 0; (w->h_dstate->c.overflow & 1) != 0; ++attempt)
	{
		if (!w->toiSnapshotTaken || attempt == 3) return setError(B2HIP_ERR_CAPACITY, "contact array full during a TOI sub-step");
		// (edits can only be pending here if a PreSolve called from a sub-step made them: toiPreSolveRounds)
		if (!w->dirtyList.empty() || !w->editOps.empty() || !w->proxyEdits.empty() || !w->pendingMoves.empty())
			return setError(B2HIP_ERR_CAPACITY, "contact array full during a TOI sub-step whose PreSolve edited the world");
		LAUNCH(w, k_toi_snapshot, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, 1);
		rc = ensureCapacity(w, 2 * (size_t)w->dw.capContacts);
		if (rc) return rc;
		rc = uploadToiVerdicts(w);
		if (rc) return rc;
		HIP_TRY(hipMemsetAsync(&w->d_state.p->c.overflow, 0, sizeof(int), w->stream));
		w->toiChains = false;
		w->toiSpeculative = false;
		rc = phaseToiSync(w);
		if (rc) return rc;
		if (w->toiChains)
		{
			// (the parallel paths report through toiUnsafe: take their serial fallback here as well)
			rc = downloadState(w, -1);
			if (rc) return rc;
			if (w->h_dstate->c.toiUnsafe != 0)
			{
				LAUNCH(w, k_toi_snapshot, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, 1);
				rc = toiSerial(w);
				if (rc) return rc;
				w->toiFallbacks += 1;
			}
		}
		rc = downloadState(w, -1);
		if (rc) return rc;
	}
	return 0;
}

// The answers collected so far go to the device before the phase runs again (DW::toiVerdict, DW::nToiVerdict).
static int uploadToiVerdicts(b2hip_world* w)
{
	const size_t n = std::min(w->toiVerdicts.size(), w->toiVerdict.cap);
	w->dw.nToiVerdict = w->dw.toiVerdict != nullptr ? (int)n : 0;
	if (w->dw.nToiVerdict > 0) HIP_TRY(hipMemcpyAsync((void*)w->toiVerdict.p, w->toiVerdicts.data(), n * sizeof(int4), hipMemcpyHostToDevice, w->stream));
	return 0;
}

static void toiCallbackFromLog(const ToiLogRec& r, b2hip_toi_callback* cb)
{
	memset(cb, 0, sizeof(*cb));
	cb->kind = r.info.x;
	cb->contact_index = r.info.y;
	cb->fixture_a = r.info.z;
	cb->fixture_b = r.info.w;
	toManifold(&cb->old_manifold, r.o0, r.o1, r.oimp, r.o3);
	toManifold(&cb->manifold, r.n0, r.n1, r.nimp, r.n3);
	cb->material.friction = r.mat.x;
	cb->material.restitution = r.mat.y;
	cb->material.tangent_speed = r.mat.z;
}

// b2ContactListener::PreSolve from INSIDE the TOI sub-steps (b2World.cpp:866,946 -> b2Contact::Update -> b2Contact.cpp:283-297).
// The reference calls it in the middle of its event loop, and what it does to the contact changes that sub-step: a contact
// switched off keeps the sweeps of its bodies and stays out of the sub-step's island (b2World.cpp:873-881, 948-954), an
// edited material is what the sub-step's solver reads. The event loop here is one kernel, so the phase is run to its end,
// its log (DW::toiLog: the Updates in the reference's call order) read, and PreSolve called for the logged Updates in
// order - each exactly once. An answer that changes nothing needs nothing: the contact stays on (or off, where the sub-step
// assumed the listener's last answer for this contact: CF_PRESOLVE_OFF, toiPreSolveOutcome) and its material as it was. The
// first one that does makes everything after it void: the phase goes back to its snapshot and runs again with the
// answers so far on the device (the loop applies them at the same log slots - it is deterministic, so the log repeats
// itself up to there), and the calls go on behind the slot that was answered. One extra run of the phase per changing answer.
static int toiPreSolveRounds(b2hip_world* w, std::vector<ToiLogRec>& recs)
{
	int asked = 0; // log slots whose PreSolve has been called
	for (int round = 0;; ++round)
	{
		const int n = std::min(w->h_dstate->c.nToiLog, w->dw.capToiLog);
		recs.resize((size_t)std::max(n, 0));
		if (n > 0) HIP_TRY(hipMemcpy((void*)recs.data(), w->toiLog.p, (size_t)n * sizeof(ToiLogRec), hipMemcpyDeviceToHost));
		if (!hasPreSolve(w) || w->dw.toiVerdict == nullptr) return 0;
		// (the log itself was cut short: no listener call from a truncated log - the step fails with the capacity error below)
		if (w->h_dstate->c.toiOverflow & 64) return 0;
		// The callbacks below may edit bodies (callbackWindow). h_state holds the state AFTER this step's phases by now, but the
		// mirror's epoch is only advanced at the very end of the step (refreshMirror): a body the host had touched before the
		// step (a force applied every frame) would not be pulled again and the edit would land on - and later upload - its
		// pre-step row. The read-back that just happened is the mirror from here on.
		if (n > asked) refreshMirror(w);
		bool again = false;
		for (int k = asked; k < n && !again; ++k)
		{
			const ToiLogRec& r = recs[(size_t)k];
			int4 v = make_int4(0, 0, 0, 0);
			if (r.info.x & 4)
			{
				b2hip_toi_callback cb;
				toiCallbackFromLog(r, &cb);
				b2hip_pre_solve_record rec;
				rec.contact_index = cb.contact_index;
				rec.fixture_a = cb.fixture_a;
				rec.fixture_b = cb.fixture_b;
				rec.enabled = 1;
				rec.old_manifold = cb.old_manifold;
				rec.manifold = cb.manifold;
				rec.material = cb.material;
				// (world edits made from the callback are taken like edits between steps: they reach the device before the next step)
				w->callbackWindow = true;
				if (w->preSolveBatchFn) w->preSolveBatchFn(w->preSolveUser, 1, &rec);
				else rec.enabled = w->preSolveFn(w->preSolveUser, rec.contact_index, rec.fixture_a, rec.fixture_b, &rec.old_manifold, &rec.manifold, &rec.material) ? 1 : 0;
				w->callbackWindow = false;
				int bits[3];
				memcpy(bits, &rec.material, sizeof(bits));
				v = make_int4(1 | (rec.enabled ? 0 : 2), bits[0], bits[1], bits[2]);
				// (bit 4 of the kind: the sub-step went on as if the contact had been switched off - the listener's last answer)
				const bool assumedOff = (r.info.x & 16) != 0;
				again = (rec.enabled != 0) == assumedOff || memcmp(&rec.material.friction, &r.mat.x, 4) != 0 || memcmp(&rec.material.restitution, &r.mat.y, 4) != 0 ||
					memcmp(&rec.material.tangent_speed, &r.mat.z, 4) != 0;
			}
			w->toiVerdicts.push_back(v);
			asked = k + 1;
		}
		if (!again) return 0;
		if (!w->toiSnapshotTaken) return setError(B2HIP_ERR_INVALID, "PreSolve changed a contact inside a TOI sub-step, and the phase kept no snapshot");
		if (round >= w->dw.capToiLog) return setError(B2HIP_ERR_INVALID, "TOI PreSolve rounds do not end");
		// (the host mirror of an edited body was refreshed from the state that is about to be taken back)
		if (!w->dirtyList.empty() || !w->editOps.empty() || !w->proxyEdits.empty() || !w->pendingMoves.empty())
			return setError(B2HIP_ERR_INVALID, "a PreSolve inside a TOI sub-step edited the world AND changed its contact: not supported");
		LAUNCH(w, k_toi_snapshot, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, 1);
		int rc = uploadToiVerdicts(w);
		if (rc) return rc;
		w->toiChains = false;
		w->toiSpeculative = false;
		rc = phaseToiSync(w);
		if (rc) return rc;
		rc = downloadState(w, -1);
		if (rc) return rc;
		rc = settleToiOverflow(w);
		if (rc) return rc;
		w->toiPreSolveReruns += 1;
	}
}

static int stepEndImpl(b2hip_world* w)
{
	int rc = downloadState(w, -1, w->sp.dt > 0.0f); // (rows only if this is the last read-back of the step: k_end_step)
	if (rc) return rc;
	// optimistic small-sort path overflowed (or the pair buffer itself): finish the pair update with the radix path (after
	// growing the buffer and searching again), then read back again
	// bit 1: candidate-pair buffer; bit 0 with moves still buffered: the new contacts did not fit and creation was skipped
	// as a whole (createBlocked) - both are cured by growing and running the pair update again. Bit 0 without buffered
	// moves comes from a contact created inside a TOI sub-step: that one is lost.
	if (w->h_dstate->c.overflow & 16) return setError(B2HIP_ERR_CAPACITY, "more TOI-candidate contacts destroyed in one step than the pair buffer has room to order (toiOrderDestroy)");
	const bool pairOverflow = (w->h_dstate->c.overflow & 2) != 0 || ((w->h_dstate->c.overflow & 1) != 0 && w->h_dstate->c.nMoves != 0);
	if ((w->h_dstate->c.overflow & 1) != 0 && !pairOverflow) return setError(B2HIP_ERR_CAPACITY, "contact array full during a TOI sub-step");
	if (pairOverflow && w->sp.dt <= 0.0f) return setError(B2HIP_ERR_CAPACITY, "pair buffer overflow");
	if ((w->h_dstate->c.nMoves != 0 || pairOverflow) && w->sp.dt > 0.0f && w->stepSolves)
	{
		if (w->h_dstate->c.nPairs > COUNT_RANK_MAX || pairOverflow)
		{
			w->pairsLargeSticky = 16; // (the next steps ask for the pair count right after the search: findNewContactsGraph)
			const bool redoToi = w->toiSpeculative;
			if (redoToi && w->h_dstate->c.nToiList > 0)
			{
				// the TOI phase ran without the contacts that are created only now: undo it
				LAUNCH(w, k_toi_snapshot, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, 1);
			}
			if (pairOverflow)
			{
				rc = growPairBuffers(w);
				if (rc) return rc;
				rc = findNewContacts(w, true);
			}
			else rc = runSortAndCreate(w, true, w->h_dstate->c.nPairs);
			if (rc) return rc;
			if (redoToi)
			{
				w->toiChains = false;
				w->toiSpeculative = false;
				// (the redone phase is judged on its own: not by what the speculative launch assumed of the grid - ADVICE round 5)
				w->toiSpecDomains = false;
				w->toiSpecGridAssumed = false;
				rc = phaseToiSync(w);
				if (rc) return rc;
			}
			rc = downloadState(w, -1);
			if (rc) return rc;
		}
	}
	if (w->toiSpeculative)
	{
		const Counters& tc = w->h_dstate->c;
		if (tc.nToiList > w->dw.capContacts) return setError(B2HIP_ERR_CAPACITY, "TOI list overflow");
		w->last.nToiList = tc.nToiList;
		w->last.nToiCalls = tc.nToiCalls;
		w->toiRan = tc.nToiList > 0;
		if (tc.nToiList == 0) w->toiChains = false;
	}
	if (w->toiSpecDomains)
	{
		// the component path was queued without its census (phaseToi): the hints for the next step, and the one assumption
		w->toiSpecDomains = false;
		w->lastToiList = w->h_dstate->c.nToiList;
		w->gridFreshLast = w->h_dstate->c.gridFresh != 0;
		if (w->h_dstate->c.nToiList > 0) w->toiDomainsSticky = 16;
		else if (w->toiDomainsSticky > 0) w->toiDomainsSticky -= 1;
		// (the event loops searched a grid that this step's pair update had NOT rebuilt - nothing moved, the first step in
		// many: as with any order-dependent case, back to the snapshot and the serial loop, which builds its own)
		if (w->toiChains && w->toiSpecGridAssumed && w->h_dstate->c.gridFresh == 0) w->h_dstate->c.toiUnsafe |= 0x80;
	}
	if (w->toiChains)
	{
		if (w->h_dstate->c.nToiMoved > 0) w->toiGridSticky = 16;
		else if (w->toiGridSticky > 0) w->toiGridSticky -= 1;
	}
	if (w->toiChains && w->h_dstate->c.toiUnsafe != 0)
	{
		// a chain met an order-dependent case: back to the state before the chains, then the reference's serial order
		LAUNCH(w, k_toi_snapshot, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, 1);
		if (w->h_dstate->c.toiUnsafe == 4 /* TOI_UNSAFE_PAIR */ && !w->toiChainsHadGrid && !w->dw.noChainCreate)
		{
			// ... unless all that happened is that a chain moved a proxy out of its fat AABB while the hash grid was not kept
			// up (nothing had moved for 16 steps): the chains once more, with the grid - most such moves find nothing, or a
			// pair the chains' close-out can create itself. (The serial loop costs ~50 us per event: 15 ms for the 290 resting
			// impacts of a 50 000-box pyramid; this costs a second first pass.)
			w->toiGridSticky = 16;
			w->toiChains = false;
			w->toiSpeculative = false;
			w->toiSpecDomains = false;
			w->toiSpecGridAssumed = false;
			rc = phaseToiSync(w);
			if (rc) return rc;
			rc = downloadState(w, -1);
			if (rc) return rc;
			w->toiGridRetries += 1;
			if (w->toiChains && w->h_dstate->c.toiUnsafe != 0) LAUNCH(w, k_toi_snapshot, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, 1);
		}
	}
	if (w->toiChains && w->h_dstate->c.toiUnsafe != 0)
	{
		if (getenv("B2HIP_TOI_WHY")) fprintf(stderr, "b2hip: TOI fallback to the serial loop, unsafe bits 0x%x (1 partner, 2 woke, 4 new pair, 8 capacity, 16 moved proxies), %d pending, %d components\n", w->h_dstate->c.toiUnsafe, w->h_dstate->c.nToiList, w->h_dstate->c.nToiDomains);
		// (a capacity cut - possibly more candidate contacts in one event than the narrow component loops have lanes: the wide form next time)
		if (w->h_dstate->c.toiUnsafe & 8) w->toiDomWide = 64;
		rc = toiSerial(w);
		if (rc) return rc;
		w->toiFallbacks += 1;
		w->toiSyncSticky = 16;
		rc = downloadState(w, -1);
		if (rc) return rc;
	}
	rc = settleToiOverflow(w);
	if (rc) return rc;
	w->postSolve.clear();
	if (w->postSolveOn)
	{
		const int n = std::min(w->h_dstate->c.nPostSolve, w->dw.capContacts);
		if (n > 0)
		{
			std::vector<PostSolveRec> recs(n);
			HIP_TRY(hipMemcpy((void*)recs.data(), w->postRecs.p, (size_t)n * sizeof(PostSolveRec), hipMemcpyDeviceToHost));
			std::vector<std::pair<unsigned long long, int> > order(n);
			for (int i = 0; i < n; ++i) order[i] = std::make_pair(recs[i].key, i);
			std::sort(order.begin(), order.end(), keyLess); // b2DeferredPostSolveLessThan: proxy-id pair
			w->postSolve.resize(n);
			for (int k = 0; k < n; ++k)
			{
				const PostSolveRec& r = recs[order[k].second];
				b2hip_contact_impulse& o = w->postSolve[k];
				o.contact_index = r.info.x;
				o.fixture_a = r.info.y;
				o.fixture_b = r.info.z;
				o.count = r.info.w;
				o.normal_impulses[0] = r.imp.x; o.tangent_impulses[0] = r.imp.y;
				o.normal_impulses[1] = r.imp.z; o.tangent_impulses[1] = r.imp.w;
			}
		}
	}
	w->toiCallbacks.clear();
	if (w->dw.toiLog != nullptr)
	{
		std::vector<ToiLogRec> recs;
		rc = toiPreSolveRounds(w, recs);
		if (rc) return rc;
		for (size_t k = 0; k < recs.size(); ++k)
		{
			b2hip_toi_callback cb;
			toiCallbackFromLog(recs[k], &cb);
			cb.kind &= ~(4 | 16); // (PreSolve has been called: toiPreSolveRounds)
			if (cb.kind == 0) continue; // (an Update that called nothing else: the contact neither began nor ended)
			w->toiCallbacks.push_back(cb);
		}
	}
	w->events.clear();
	// (the rows were left out of a read-back because another one was due, and it did not come: safety net, never seen)
	if (w->h_dstate->c.rowsSkipped == 1)
	{
		rc = downloadState(w, -1);
		if (rc) return rc;
	}
	if (w->eventsOn)
	{
		// after every fallback has had its say: one pass over the contacts, then the (usually short) list comes back
		LAUNCH(w, k_contact_events, gridFor(w->dw.capContacts), 256, w->dw);
		int nEv = 0;
		HIP_TRY(hipMemcpyAsync(&nEv, &w->d_state.p->c.nEvents, sizeof(int), hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
		if (nEv > w->dw.capContacts) return setError(B2HIP_ERR_CAPACITY, "contact event buffer overflow");
		if (nEv > 0)
		{
			std::vector<unsigned long long> keys(nEv);
			std::vector<int4> info(nEv);
			HIP_TRY(hipMemcpy(keys.data(), w->evKey.p, nEv * sizeof(unsigned long long), hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(info.data(), w->evInfo.p, nEv * sizeof(int4), hipMemcpyDeviceToHost));
			std::vector<int> order(nEv);
			for (int i = 0; i < nEv; ++i) order[i] = i;
			// begins before ends, each group by proxy-id pair (b2ContactManager.cpp:420-438, b2ContactPointerLessThan :64-67)
			std::sort(order.begin(), order.end(), [&](int a, int b)
			{
				if (info[a].z != info[b].z) return info[a].z < info[b].z;
				return keys[a] < keys[b];
			});
			w->events.resize(nEv);
			for (int i = 0; i < nEv; ++i)
			{
				const int4 q = info[order[i]];
				w->events[i].fixture_a = q.x;
				w->events[i].fixture_b = q.y;
				w->events[i].kind = q.z;
				w->events[i].contact_index = q.w;
			}
		}
	}
	// (no synchronisation here: the read-back has arrived - awaitState - and k_end_step was the last thing on the stream)
	if (w->debugSync) HIP_TRY(hipStreamSynchronize(w->stream));
	refreshMirror(w);
	const Counters& c = w->h_dstate->c;
	w->lastContacts = c.nContacts;
	w->last.nContacts = c.nContacts;
	w->last.nMoves = c.nMoves;
	if (c.nMovesSeen > 256 && !w->gridForced)
	{
		// the grid's cell for the next step, from what this step's pair search went through (b2d_kernels_broadphase.h: gridCell)
		long long rounds = 0;
		for (int k = 0; k < 32; ++k) rounds += c.candRounds[k];
		const double perProxy = 64.0 * (double)rounds / (double)c.nMovesSeen;
		if (perProxy > 128.0) w->gridHalf = true; else if (perProxy < 48.0) w->gridHalf = false;
		w->dw.gridHalf = w->gridHalf ? 1 : 0; // (between steps: every kernel of the next step sees the same geometry)
	}
	w->last.nNewContacts = c.nNewContacts;
	w->last.nPairs = c.nPairs;
	w->last.overflow = c.overflow;
	if (w->colorSmallPending)
	{
		w->colorSmallPending = false;
		w->last.nColors = c.nColors;
		if (c.overflow & 4) return setError(B2HIP_ERR_CAPACITY, "more than 64 constraint colours on one body");
		if (c.nUncolored != 0) return setError(B2HIP_ERR_CAPACITY, "incremental colouring did not converge");
	}
	if ((c.overflow & 0x2040) == 0x2040) return setError(B2HIP_ERR_CAPACITY, "a block of the large-island partition holds more rows or home bodies than its workgroup takes (the census the host partitions by did not see them)");
	if (c.overflow & 64)
	{
		if (getenv("B2HIP_HANDOVER_WHY")) handoverPostMortem(w);
		return setError(B2HIP_ERR_HIP, "a wait inside a resident large-island solver timed out (a workgroup was not resident, or a hand-over never came)");
	}
	if (c.overflow & SCAN_ABORT_BIT) return setError(B2HIP_ERR_HIP, "a single-pass scan gave up waiting for a predecessor tile (k_scan_chain look-back)");
	w->last.posItersLarge = c.posItersLarge;
	w->last.nHubRows = c.nHubRows;
	w->last.hubRounds = c.hubRounds;
	w->last.hubSerialChunks = c.hubSerialChunks;
	w->toiChainContacts += c.nToiChainCreated;
	if (w->toiRan)
	{
		w->last.toiUnsafe = c.toiUnsafe;
		w->last.nToiEvents = c.nToiEvents;
		w->last.nToiCalls = c.nToiCalls;
		w->last.toiOverflow = c.toiOverflow;
		if (c.toiOverflow) return setError(B2HIP_ERR_CAPACITY, "TOI event scratch overflow (flags " + std::to_string(c.toiOverflow) + ")");
	}
	if (w->sp.dt > 0.0f) w->inv_dt0 = w->sp.inv_dt;
	// b2World::m_stepComplete (b2World.cpp:1072, 1084): only SolveTOI changes it
	if (w->def.continuous && w->sp.dt > 0.0f) w->stepComplete = c.toiIncomplete == 0;
	w->stepActive = false;

	// b2Profile from events (milliseconds)
	float ms = 0.0f;
	float* p = w->profile;
	memset(p, 0, sizeof(float) * 13);
	{
		const unsigned long long* pc0 = w->h_dstate->phaseClock;
		p[0] = pc0[13] > pc0[14] ? 1.0e-5f * (float)(pc0[13] - pc0[14]) : 0.0f;      // step (device clock, read-back included)
	}
	if (w->profileDetail)
	{
	// the other figures: device clock (10 ns ticks) at the start of the first kernel of each phase (stampPhase)
	const unsigned long long* pc = w->h_dstate->phaseClock;
	auto span = [pc](int a, int b) -> float { return pc[b] > pc[a] ? 1.0e-5f * (float)(pc[b] - pc[a]) : 0.0f; };
	p[1] = span(1, 2);                                                           // collide
	p[2] = span(2, 3);                                                           // solve (islands + solver)
	p[3] = span(2, 4);                                                           // solveTraversal = island build
	const float dfs = span(4, 5), small = span(5, 6), color = span(6, 7), large = span(7, 8);
	p[3] += dfs + color;
	p[5] = small + large;                                                        // solver kernels (init+velocity+position)
	const float bpTop = span(0, 1), bp0 = span(3, 9), bp1 = span(9, 10);
	p[10] = bp0;                                                                 // broadphaseSyncFixtures
	p[11] = bp1 + bpTop;                                                         // broadphaseFindContacts
	p[9] = bp0 + bp1 + bpTop;                                                    // broadphase
	if (w->toiEventValid) p[7] = span(10, 12);                                   // solveTOI
	w->solverMs = small + large;
	if (w->blocksThisStep && w->h_dstate->stamps[4] > 0)
	{
		// b2Profile::solveInit / solveVelocity / solvePosition (b2TimeStep.h:30-32) from the block solver's own phase stamps:
		// [0] constraints initialised, [1] velocity iterations done, [2] positions integrated, [3] position iterations done,
		// [4] written back (10 ns ticks); scaled to the event-measured span of the launch (stamps are workgroup 0's view)
		const float tick = 1.0e-5f; // ms
		const float total = tick * (float)w->h_dstate->stamps[4];
		const float scale = total > 0.0f ? large / total : 0.0f;
		p[4] = scale * tick * (float)w->h_dstate->stamps[0];
		p[5] = small + scale * tick * (float)(w->h_dstate->stamps[2] - w->h_dstate->stamps[0]);
		p[6] = scale * tick * (float)(w->h_dstate->stamps[4] - w->h_dstate->stamps[2]);
	}
	}
	const int Ct = w->last.nSContacts + w->last.nLContacts;
	const int B = w->last.nSBodies + w->last.nLBodies;
	w->solverConstraints = Ct;
	w->solverBodies = B;
	// Ct*(Nv*220 + Np*136 + 488) + B*240 with Np = configured position iterations
	w->solverBytes = (double)Ct * (w->sp.velIters * 220.0 + w->sp.posIters * 136.0 + 488.0) + (double)B * 240.0;
	w->ktMs = 0.0f;
	w->ktLaunches = 0;
	w->ktBytes = 0.0;
	if (w->kernelTiming && w->ktUsed >= 2)
	{
		for (int k = 0; k + 1 < w->ktUsed; k += 2)
		{
			float t = 0.0f;
			(void)hipEventElapsedTime(&t, w->ktEvents[k], w->ktEvents[k + 1]);
			w->ktMs += t;
			w->ktLaunches += 1;
		}
		// SURVEY.md 8d per-unit figures: collide 480 B per contact of two polygons (230 B otherwise: circles), sync fixtures
		// 250 B per proxy, pair update 16 B per proxy read + 8 B per candidate pair written
		if (w->ktKind == 5) w->ktBytes = (double)w->ktUnitsA * 480.0 + (double)w->ktUnitsB * 230.0;
		else if (w->ktKind == 6) w->ktBytes = (double)w->ktUnitsA * 250.0;
		else if (w->ktKind == 7) w->ktBytes = (double)w->ktUnitsA * 16.0 + (double)w->ktUnitsB * 8.0;
		else if (w->ktKind == 1) w->ktBytes = (double)w->last.nLContacts * 220.0 * w->sp.velIters;
		else if (w->ktKind == 8)
		{
			// the whole family: SURVEY 8d's solver figure for the large islands, position iterations as executed
			w->ktBytes = (double)w->last.nLContacts * (w->sp.velIters * 220.0 + w->last.posItersLarge * 136.0 + 488.0) + (double)w->last.nLBodies * 240.0;
			w->ktLaunches = w->familyLaunches;
		}
		else if (w->ktKind == 3 || w->ktKind == 4) w->ktBytes = (double)w->last.nLContacts * (w->sp.velIters * 220.0 + w->last.posItersLarge * 136.0 + 488.0) + (double)w->last.nLBodies * 240.0;
		else w->ktBytes = (double)w->last.nSContacts * (w->sp.velIters * 220.0 + w->sp.posIters * 136.0 + 488.0) + (double)w->last.nSBodies * 240.0;
	}
	return 0;
}

int b2hip_set_lazy_readback(b2hip_world* w, int enable)
{
	if (int rc = checkUsable(w, "b2hip_set_lazy_readback", true)) return rc;
	ensureRows(w); // (switching off with rows outstanding: they come home now)
	w->lazyReadback = enable != 0;
	return w->failed ? setError(B2HIP_ERR_HIP, w->failedWhy) : B2HIP_OK;
}

int b2hip_step_end(b2hip_world* w)
{
	if (int rc = checkUsable(w, "b2hip_step_end", false)) return rc;
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_step_end outside a step");
	DEVICE_GUARD(w);
	const int rc = stepFailed(w, stepEndImpl(w));
	if (w->traceLaunches) { fprintf(stderr, "[b2hip] host: step end returns %d\n", rc); fflush(stderr); }
	return rc;
}

int b2hip_step(b2hip_world* w, float dt, int velocity_iterations, int position_iterations)
{
	int rc = b2hip_step_begin(w, dt, velocity_iterations, position_iterations);
	if (rc) return rc;
	rc = b2hip_collide(w);
	if (rc) return rc;
	rc = b2hip_solve(w);
	if (rc) return rc;
	rc = b2hip_sync_fixtures(w);
	if (rc) return rc;
	rc = b2hip_find_new_contacts(w);
	if (rc) return rc;
	rc = b2hip_solve_toi(w);
	if (rc) return rc;
	return b2hip_step_end(w);
}

int b2hip_get_body_states(b2hip_world* w, int first, int count, b2hip_body_state* out)
{
	if (!w || !out || first < 0 || count < 0 || first + count > (int)w->bodies.size()) return setError(B2HIP_ERR_INVALID, "bad range");
	ensureRows(w);
	if (w->failed) return setError(B2HIP_ERR_INVALID, "b2hip_get_body_states: the world is in a failed state (" + w->failedWhy + ")");
	for (int i = 0; i < count; ++i)
	{
		const HostBody& b = w->bodies[first + i];
		b2hip_body_state& s = out[i];
		if (!b.dirty && b.pullEpoch != w->mirrorEpoch && (size_t)(first + i) < w->stateCount)
		{
			// straight from the pinned read-back buffer (same 40-byte layout)
			memcpy(&s, w->h_state + 10 * (size_t)(first + i), sizeof(b2hip_body_state));
			s.flags = (s.flags & 0x7cu) | (uint32_t)b.type;
			continue;
		}
		s.px = b.px; s.py = b.py; s.angle = b.a;
		s.vx = b.vx; s.vy = b.vy; s.w = b.w;
		s.cx = b.cx; s.cy = b.cy;
		s.flags = (b.flags & 0x7cu) | (uint32_t)b.type;
		s.sleep_time = b.sleepTime;
	}
	return 0;
}

// Edits queued since the last step (destroyed bodies / fixtures ...) change the contact list at once in the reference:
// whoever looks at the contacts between steps sees them applied.
static int flushForRead(b2hip_world* w)
{
	if (w->editOps.empty() || w->stepActive || w->failed) return 0;
	DEVICE_GUARD(w);
	int rc = flushEdits(w);
	if (rc) return rc;
	return applyEditOps(w, true);
}

int b2hip_contact_count(b2hip_world* w)
{
	if (!w) return 0;
	(void)flushForRead(w);
	return w->lastContacts;
}

int b2hip_enable_contact_events(b2hip_world* w, int enable)
{
	if (int rcu = checkUsable(w, "b2hip_enable_contact_events", true)) return rcu;
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	w->eventsOn = enable != 0;
	w->dw.eventsOn = w->eventsOn ? 1 : 0;
	w->events.clear();
	return B2HIP_OK;
}

int b2hip_get_contact_events(b2hip_world* w, int cap, b2hip_contact_event* out)
{
	if (!w || (cap > 0 && !out)) return setError(B2HIP_ERR_INVALID, "null argument");
	const int n = (int)w->events.size();
	for (int i = 0; i < n && i < cap; ++i) out[i] = w->events[i];
	return n;
}

int b2hip_get_toi_callbacks(b2hip_world* w, int cap, b2hip_toi_callback* out)
{
	if (!w || (cap > 0 && !out)) return setError(B2HIP_ERR_INVALID, "null argument");
	const int n = (int)w->toiCallbacks.size();
	for (int i = 0; i < n && i < cap; ++i) out[i] = w->toiCallbacks[i];
	return n;
}

