// b2hip_api_world.h - part of the ONE translation unit b2hip.hip, inside its extern "C" block: the C ABI of include/b2hip.h that
// builds and edits a world - create / destroy of worlds, bodies, fixtures, the eleven joint types, every setter.
// (No include guard on purpose: b2hip.hip includes it exactly once, in order - the fragments share one scope.)


const char* b2hip_last_error(void)
{
	return g_lastError.c_str();
}

const char* b2hip_version(void)
{
	return "b2hip 0.1 (gfx950)";
}

int b2hip_world_create(const b2hip_world_def* def, b2hip_world** out)
{
	if (!def || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0)
	{
		return setError(B2HIP_ERR_NO_DEVICE, "no HIP device available: the b2hip Step() path has no CPU fallback");
	}
	int current = 0;
	if (hipGetDevice(&current) != hipSuccess) current = 0;
	const int device = def->device >= 0 ? def->device : current;
	if (device >= count) return setError(B2HIP_ERR_NO_DEVICE, "no such HIP device");
	b2hip_world* w = new b2hip_world();
	w->def = *def;
	w->device = device; // the ordinal itself, also when the caller asked for "current": later calls select it again
	DEVICE_GUARD(w);
	{
		int now = -1;
		if (hipGetDevice(&now) != hipSuccess || now != device)
		{
			delete w;
			return setError(B2HIP_ERR_NO_DEVICE, "hipSetDevice failed");
		}
	}
	e = hipStreamCreateWithFlags(&w->stream2, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&w->evFork, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&w->evJoin, hipEventDisableTiming);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
	if (e != hipSuccess)
	{
		delete w;
		return setError(B2HIP_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
	}
	w->debugSync = getenv("B2HIP_DEBUG") != nullptr;
	w->forceLarge = getenv("B2HIP_FORCE_LARGE") ? atoi(getenv("B2HIP_FORCE_LARGE")) : 0;
	w->nextNode = 0;
	w->leafCount = 0;
	w->upBodies = w->upFixtures = w->upShapes = w->upJoints = 0;
	w->stateCount = 0;
	w->mirrorEpoch = 1;
	w->stepEpoch = 1;
	w->newFixture = false;
	w->inv_dt0 = 0.0f;
	w->stepActive = false;
	w->callbackWindow = false;
	w->h_state = nullptr;
	w->h_stateCap = 0;
	w->lastContacts = 0;
	memset(&w->last, 0, sizeof(w->last));
	memset(&w->dw, 0, sizeof(w->dw));
	memset(w->profile, 0, sizeof(w->profile));
	w->solverMs = 0.0f;
	w->solverBytes = 0.0;
	w->solverConstraints = w->solverBodies = 0;
	w->kernelTiming = 0;
	w->ktUnitsA = w->ktUnitsB = 0;
	w->ktUsed = 0;
	w->ktKind = 0;
	w->ktMs = 0.0f;
	w->ktLaunches = 0;
	w->ktBytes = 0.0;
	w->dw.cellSize = 1.0f;
	w->dw.invCellSize = 1.0f;
	w->toiRan = false;
	w->toiEventValid = false;
	w->debugTrace = getenv("B2HIP_TRACE") != nullptr;
	w->kernelTimingLaunches = getenv("B2HIP_SOLVER_LAUNCHES") != nullptr; // force the launch-per-colour solver
	w->solverBarriers = getenv("B2HIP_SOLVER_BARRIERS") != nullptr;       // persistent kernel with a grid barrier per colour instead of body-level dataflow
	w->persistSteps = 0;
	w->colorSmallPending = false;
	w->hubSteps = 0;
	w->useGraphs = getenv("B2HIP_GRAPHS") != nullptr; // opt-in: measured no gain on MI355X (the step is not host-bound), see DESIGN.md
	w->graphCaptures = 0;
	w->persistMaxWG = 0;
	w->nCU = 256;
	{
		hipDeviceProp_t prop;
		int devId = 0;
		if (hipGetDevice(&devId) == hipSuccess && hipGetDeviceProperties(&prop, devId) == hipSuccess)
		{
			w->nCU = prop.multiProcessorCount;
#if B2HIP_HAVE_VALIDATION_SOLVERS
			int perCU = 0, perCU2 = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_solve_dataflow, PERSIST_LANES, 0) == hipSuccess &&
				hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU2, k_solve_mailbox<true>, PERSIST_LANES, 0) == hipSuccess)
			{
				perCU = std::min(perCU, perCU2);
				// the occupancy query can be one block per CU high (sgpr_count 81-112, MI355X_MICROARCH.md): keep a margin
				w->persistMaxWG = std::max(0, std::min(perCU - 1, 4)) * prop.multiProcessorCount;
			}
#else
			w->persistMaxWG = 1 << 24; // (a limit of the cross-check solvers; k_solve_blocks has its own: blocksMaxWG)
#endif
		}
	}
	w->dfLanesForced = 0;
	w->noSideStream = getenv("B2HIP_NO_SIDE_STREAM") != nullptr;
	w->profileDetail = !(getenv("B2HIP_PROFILE_DETAIL") && atoi(getenv("B2HIP_PROFILE_DETAIL")) == 0);
	if (const char* e = getenv("B2HIP_SOLID_ROUNDS")) { const int v = atoi(e); w->solidRoundsEnv = (v == 1 || v == 2 || v == 4 || v == 8) ? v : 0; }
	w->collideSplitEnv = getenv("B2HIP_COLLIDE_SPLIT") ? atoi(getenv("B2HIP_COLLIDE_SPLIT")) : -1;
	w->collideUniOff = getenv("B2HIP_COLLIDE_UNI") && atoi(getenv("B2HIP_COLLIDE_UNI")) == 0;
	w->collideSortEnv = getenv("B2HIP_COLLIDE_SORT") ? atoi(getenv("B2HIP_COLLIDE_SORT")) : -1;
	w->collideStage = getenv("B2HIP_COLLIDE_STAGE") ? atoi(getenv("B2HIP_COLLIDE_STAGE")) : -1;
	w->dfEpoch = 0;
	// single-XCD attempt of k_solve_mailbox: opt-in. Measured on the 10k-body pyramid it LOSES (launch 500 us against 368):
	// 334 waves polling on 32 CUs load the consumer CUs' memory queues, which is where a hand-off is priced; L2 locality
	// buys only 0.1-0.3 us of it (MI355X_MICROARCH.md, handoff-1to1)
	w->solverLocal = getenv("B2HIP_SOLVER_SINGLE_XCD") != nullptr;
	w->solverRows = getenv("B2HIP_SOLVER_ROWS") != nullptr; // polled body rows (k_solve_dataflow) instead of pushed mailboxes
	w->solverMailbox = getenv("B2HIP_SOLVER_MAILBOX") != nullptr; // pushed hand-offs for every constraint (k_solve_mailbox) instead of k_solve_blocks
	w->noSweepBlocks = getenv("B2HIP_NO_SWEEP_BLOCKS") != nullptr;
	w->tracePartition = getenv("B2HIP_TRACE_PARTITION") != nullptr;
	if (const char* e = getenv("B2HIP_GRID_HALF")) { w->gridForced = true; w->gridHalf = atoi(e) != 0; w->dw.gridHalf = w->gridHalf ? 1 : 0; }
	w->dw.noChainCreate = getenv("B2HIP_TOI_NO_CHAIN_CREATE") != nullptr ? 1 : 0;
	w->dw.noOwnIdBlocks = getenv("B2HIP_NO_OWN_ID_BLOCKS") != nullptr ? 1 : getenv("B2HIP_OWN_ID_BLOCKS_ALL") != nullptr ? -1 : 0;
	w->traceLaunches = getenv("B2HIP_TRACE_LAUNCHES") != nullptr;
	w->noBlocks = getenv("B2HIP_NO_BLOCKS") != nullptr;           // no block partition at all (colours as before it existed)
	w->blockLanes = 0; // chosen per partition (see phaseSolve); B2HIP_BLOCK_LANES = 256 | 512 | 1024 forces one size
	if (const char* e = getenv("B2HIP_BLOCK_LANES")) w->blockLanes = atoi(e) == 512 ? 512 : (atoi(e) == 256 ? 256 : (atoi(e) == 1024 ? 1024 : 0));
	{
		int perCU = 0, perCU2 = 0, perCU3 = 0;
		hipDeviceProp_t prop;
		int devId = 0;
		if (hipGetDevice(&devId) == hipSuccess && hipGetDeviceProperties(&prop, devId) == hipSuccess &&
			hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_solve_blocks<256>, 256, 0) == hipSuccess &&
			hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU2, k_solve_blocks<512>, 512, 0) == hipSuccess &&
			hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU3, k_solve_blocks<BLOCK_LANES>, BLOCK_LANES, 0) == hipSuccess &&
			perCU > 0 && perCU2 > 0 && perCU3 > 0)
		{
			// one block per CU is all this sizing relies on (the occupancy query can be one too high, MI355X_MICROARCH.md)
			w->blocksMaxWG = prop.multiProcessorCount - 8;
			int s0 = 0, s1 = 0, s2 = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&s0, k_blocks_sweep<256>, 256, 0) == hipSuccess &&
				hipOccupancyMaxActiveBlocksPerMultiprocessor(&s1, k_blocks_sweep<512>, 512, 0) == hipSuccess &&
				hipOccupancyMaxActiveBlocksPerMultiprocessor(&s2, k_blocks_sweep<BLOCK_LANES>, BLOCK_LANES, 0) == hipSuccess)
			{
				// (the blocks of a launch wait for one another: all of them must be resident together)
				w->sweepMaxWG[0] = std::max(1, s0 - 1) * w->blocksMaxWG;
				w->sweepMaxWG[1] = std::max(1, s1 - 1) * w->blocksMaxWG;
				w->sweepMaxWG[2] = std::max(1, s2 - 1) * w->blocksMaxWG;
			}
		}
	}
	{
		// the workgroups of k_large_rest / k_rest_hub wait for one another's rows: all of a launch must be resident together
		// (the grid was sized from the colour census alone)
		int devId = 0, r1 = 0, r2 = 0, f1 = 0, f2 = 0;
		hipDeviceProp_t prop;
		if (hipGetDevice(&devId) == hipSuccess && hipGetDeviceProperties(&prop, devId) == hipSuccess)
		{
			const int cus = std::max(1, prop.multiProcessorCount - 8);
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&r1, k_large_rest<1>, 256, 0) == hipSuccess &&
				hipOccupancyMaxActiveBlocksPerMultiprocessor(&r2, k_large_rest<2>, 256, 0) == hipSuccess && r1 > 0 && r2 > 0)
				w->restMaxWG = std::min(r1, r2) * cus;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&f1, k_rest_hub<1>, SWEEP_END_LANES, 0) == hipSuccess &&
				hipOccupancyMaxActiveBlocksPerMultiprocessor(&f2, k_rest_hub<2>, SWEEP_END_LANES, 0) == hipSuccess && f1 > 0 && f2 > 0)
				w->restHubMaxWG = std::min(f1, f2) * cus;
		}
	}
	w->dfSleep = 1;
	if (const char* e = getenv("B2HIP_DF_LANES")) w->dfLanesForced = std::max(64, std::min(256, atoi(e) / 64 * 64));
	if (const char* e = getenv("B2HIP_DF_SLEEP")) w->dfSleep = atoi(e);
	w->toiChains = false;
	w->toiSerialOnly = getenv("B2HIP_TOI_SERIAL") != nullptr;
	w->toiSyncOnly = getenv("B2HIP_TOI_SYNC") != nullptr;
	w->toiDomWideOnly = getenv("B2HIP_TOI_DOM_WIDE") != nullptr;
	w->noToiSpecDomains = getenv("B2HIP_TOI_NO_SPEC_DOMAINS") != nullptr;
	w->debugAssumeFreshGrid = getenv("B2HIP_DEBUG_ASSUME_FRESH_GRID") != nullptr;
	w->toiNoDomains = getenv("B2HIP_TOI_NO_DOMAINS") != nullptr; // bullets / kinematic partners through the serial loop only // decide chains / serial loop from a read-back after k_toi_first (the older flow)
	w->toiFallbacks = 0;
	for (int i = 0; i < 13; ++i) w->ev[i] = nullptr;
	w->h_dstate = nullptr;
	for (int i = 0; i < 13; ++i)
	{
		if (hipEventCreate(&w->ev[i]) != hipSuccess)
		{
			b2hip_world_destroy(w);
			return setError(B2HIP_ERR_HIP, "hipEventCreate failed");
		}
	}
	if (hipHostMalloc((void**)&w->h_dstate, sizeof(DState), hipHostMallocDefault) != hipSuccess)
	{
		b2hip_world_destroy(w);
		return setError(B2HIP_ERR_HIP, "hipHostMalloc failed");
	}
	// (written by a kernel, polled by the host: mapped and coherent)
	if (hipHostMalloc((void**)&w->h_pub, sizeof(DState), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
		hipHostGetDevicePointer((void**)&w->d_pub, w->h_pub, 0) != hipSuccess)
	{
		b2hip_world_destroy(w);
		return setError(B2HIP_ERR_HIP, "hipHostMalloc (coherent) failed");
	}
	memset(w->h_pub, 0, sizeof(DState));
	if (hipHostMalloc((void**)&w->h_pub2, sizeof(DState), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
		hipHostGetDevicePointer((void**)&w->d_pub2, w->h_pub2, 0) != hipSuccess)
	{
		b2hip_world_destroy(w);
		return setError(B2HIP_ERR_HIP, "hipHostMalloc (coherent) failed");
	}
	memset(w->h_pub2, 0, sizeof(DState));
	w->recoverOn = !(getenv("B2HIP_NO_RECOVER") && atoi(getenv("B2HIP_NO_RECOVER")));
	if (hipHostMalloc((void**)&w->h_solverWord, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
		hipHostGetDevicePointer((void**)&w->d_solverWord, w->h_solverWord, 0) != hipSuccess)
	{
		b2hip_world_destroy(w);
		return setError(B2HIP_ERR_HIP, "hipHostMalloc (coherent) failed");
	}
	memset(w->h_solverWord, 0, 64);
	w->noCensusPoll = getenv("B2HIP_NO_CENSUS_POLL") && atoi(getenv("B2HIP_NO_CENSUS_POLL"));
	w->noStatePoll = getenv("B2HIP_NO_STATE_POLL") && atoi(getenv("B2HIP_NO_STATE_POLL"));
	w->lazyReadback = getenv("B2HIP_LAZY_READBACK") && atoi(getenv("B2HIP_LAZY_READBACK"));
	if (getenv("B2HIP_EARLY_ROWS_MIN")) w->earlyRowsMin = atoi(getenv("B2HIP_EARLY_ROWS_MIN"));
	int rc = ensureCapacity(w, 0);
	if (rc == 0 && hipStreamSynchronize(w->stream) != hipSuccess) rc = setError(B2HIP_ERR_HIP, "stream sync failed");
	if (rc)
	{
		const std::string why = g_lastError;
		b2hip_world_destroy(w);
		return setError(rc, why);
	}
	*out = w;
	return B2HIP_OK;
}

void b2hip_world_destroy(b2hip_world* w)
{
	if (!w) return;
	DEVICE_GUARD(w);
	if (w->stream) (void)hipStreamSynchronize(w->stream);
	if (w->rowStream)
	{
		(void)hipStreamSynchronize(w->rowStream);
		(void)hipStreamDestroy(w->rowStream);
		(void)hipEventDestroy(w->rowFork);
		(void)hipEventDestroy(w->rowJoin);
		w->rowStream = nullptr;
	}
	if (w->shardComm != nullptr && g_rcclDestroy != nullptr) g_rcclDestroy(w->shardComm);
	w->shardComm = nullptr;
	w->shardSend.release(); w->shardRecv.release();
	w->b_owner.release(); w->spNewOwner.release(); w->spAwake.release(); w->spStraddle.release(); w->spCount.release(); w->spTarget.release();
	w->spSend.release(); w->spRecv.release(); w->spTailKey.release(); w->spVirt.release();
	if (w->spHost) (void)hipHostFree(w->spHost);
	w->spHost = nullptr;
	if (w->spOwnHost) (void)hipHostFree(w->spOwnHost);
	w->spOwnHost = nullptr;
	if (w->spHdrHost) (void)hipHostFree(w->spHdrHost);
	w->spHdrHost = nullptr;
	for (size_t k = 0; k < w->spTape.size(); ++k) (void)hipFree(w->spTape[k].first);
	w->spTape.clear();
	w->d_state.release();
	w->b_pos.release(); w->b_pos0.release(); w->b_vel.release(); w->b_xf.release(); w->b_mass.release(); w->b_damp.release();
	w->b_force.release(); w->b_flags.release(); w->b_wake.release(); w->b_rowDirty.release();
	w->p_fat.release(); w->p_body.release(); w->p_shape.release(); w->p_key.release(); w->p_filter0.release(); w->p_filter1.release();
	w->p_mat.release(); w->d_shapes.release();
	for (int k = 0; k < 2; ++k)
	{
		w->c_ids[k].release(); w->c_key[k].release(); w->c_flags[k].release(); w->c_mat[k].release(); w->c_man0[k].release();
		w->c_man1[k].release(); w->c_imp[k].release(); w->c_man3[k].release(); w->c_color[k].release();
	}
	w->ht_keys.release(); w->d_joints.release(); w->d_gears.release(); w->li_ref.release();
	w->jadjStart.release(); w->jadj.release(); w->rootJointStart.release(); w->rootJointCursor.release();
	w->lj_list.release(); w->rootJointOkay.release();
	w->parent.release(); w->rootSeed.release(); w->rootBodies.release(); w->rootContacts.release(); w->rootJoints.release();
	w->rootIsland.release(); w->deg.release(); w->adjStart.release(); w->adjCursor.release(); w->adj.release(); w->adjSlot.release();
	w->rootScanIn.release(); w->rootScanOut.release();
	w->si_root.release(); w->si_bodyStart.release(); w->si_contactStart.release(); w->si_wStart.release(); w->si_maxLevel.release();
	w->si_bodies.release(); w->si_contacts.release(); w->si_level.release(); w->si_stack.release(); w->si_lastLevel.release();
	w->b_slot.release(); w->b_island.release(); w->chunkFirst.release();
	w->li_bodies.release(); w->li_contacts.release(); w->li_roots.release(); w->li_color.release(); w->colorCount.release();
	w->colorStart.release(); w->colorCursor.release(); w->li_sorted.release(); w->bodyClaim.release(); w->rootPen.release();
	w->bodyActive.release(); w->bodyRest.release(); w->b_posv.release(); w->dfRank.release(); w->dfInbox.release(); w->evKey.release(); w->evInfo.release(); w->uncolList.release(); w->compactList.release(); w->gridBar.release(); w->hubRowOf.release(); w->hubList.release(); w->hubDelta.release(); w->hubMeta.release(); w->hubFirst.release(); w->solveSnapBody.release(); w->solveSnapImp.release(); w->solveSnapCFlags.release(); w->solveSnapJoints.release(); w->solveSnapGears.release();
	w->b_proxyHead.release(); w->p_next.release(); w->toiList.release(); w->toiPos2c.release(); w->toiDestroyList.release(); w->toiNewList.release();
	w->b_toiGroup.release(); w->toiGroups.release(); w->toiGroupCount.release(); w->toiGroupList.release(); w->toiMoved.release(); w->toiNew.release(); w->toiParent.release(); w->toiDomOf.release(); w->toiDomRoot.release(); w->toiDomCount.release(); w->toiDomBase.release(); w->toiDomFill.release(); w->toiDomFailed.release(); w->toiDomEvents.release(); w->toiDomList.release(); w->toiHull.release(); w->snapBody.release(); w->snapFat.release();
	w->c_mgr[0].release(); w->c_mgr[1].release(); w->dbgPreVel.release(); w->dbgVel.release(); w->dbgLi.release();
	w->rootSleepMin.release(); w->bodyColorMask.release(); w->rootDone.release(); w->lc.release(); w->warmDelta.release();
	w->moveBuf.release(); w->gridCount.release(); w->gridStart.release(); w->gridCursor.release(); w->gridItems.release(); w->gridFat.release(); w->arriveTree.release();
	w->largeProxies.release(); w->largeMoves.release(); w->pairKey.release(); w->pairKey2.release(); w->pairProxy.release(); w->pairProxy2.release();
	w->filterPairs.release();
	w->d_editOps.release();
	w->b_order.release(); w->orderBody.release(); w->bigRoots.release();
	w->pre_o0.release(); w->pre_o1.release(); w->pre_oimp.release(); w->pre_o3.release(); w->preRecs.release();
	w->postRecs.release(); w->filterList.release(); w->hostList.release(); w->toiLog.release(); w->toiVerdict.release();
	w->b_blk1.release(); w->b_adopt.release(); w->b_adoptStage.release(); w->blkRows.release(); w->blkRowStart.release(); w->blkCursor.release(); w->blkBodyCount.release(); w->blkBodyCursor.release();
	w->blkBodyStart.release(); w->blkBodies.release(); w->rowColor.release(); w->b_cutv.release();
	w->pairFirst.release(); w->pairRank.release(); w->scanTmp.release(); w->radixHist.release(); w->radixHistScan.release();
	w->keepFlag.release(); w->keepScan.release(); w->scanTmp4.release(); w->scanFlags.release(); w->stateOut.release(); w->consts.release();
	if (w->h_state) (void)hipHostFree(w->h_state);
	if (w->h_dstate) (void)hipHostFree(w->h_dstate);
	if (w->h_pub) (void)hipHostFree(w->h_pub);
	if (w->h_pub2) (void)hipHostFree(w->h_pub2);
	if (w->h_solverWord) (void)hipHostFree(w->h_solverWord);
	for (int i = 0; i < 13; ++i)
		if (w->ev[i]) (void)hipEventDestroy(w->ev[i]);
	for (size_t i = 0; i < w->ktEvents.size(); ++i) (void)hipEventDestroy(w->ktEvents[i]);
	{
		GraphSeg* segs[3] = { &w->segCollide, &w->segIslands, &w->segPairs };
		for (int i = 0; i < 3; ++i)
		{
			if (segs[i]->exec) (void)hipGraphExecDestroy(segs[i]->exec);
			if (segs[i]->graph) (void)hipGraphDestroy(segs[i]->graph);
		}
	}
	if (w->stream) (void)hipStreamDestroy(w->stream);
	if (w->stream2) (void)hipStreamDestroy(w->stream2);
	if (w->evFork) (void)hipEventDestroy(w->evFork);
	if (w->evJoin) (void)hipEventDestroy(w->evJoin);
	delete w;
}

int b2hip_set_gravity(b2hip_world* w, float gx, float gy)
{
	if (int rcu = checkUsable(w, "b2hip_set_gravity", true)) return rcu;
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	w->def.gravity_x = gx;
	w->def.gravity_y = gy;
	return 0;
}

int b2hip_set_flags(b2hip_world* w, int allow_sleep, int warm_starting, int continuous, int sub_stepping)
{
	if (int rcu = checkUsable(w, "b2hip_set_flags", true)) return rcu;
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	w->def.allow_sleep = allow_sleep;
	w->def.warm_starting = warm_starting;
	w->def.continuous = continuous;
	w->def.sub_stepping = sub_stepping;
	return 0;
}

int b2hip_create_body(b2hip_world* w, const b2hip_body_def* def)
{
	if (!w || !def) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rcu = checkUsable(w, "b2hip_create_body", true)) return rcu;
	HostBody b{};
	b.type = def->type;
	b.flags = 0;
	if (def->bullet) b.flags |= BF_BULLET;
	if (def->fixed_rotation) b.flags |= BF_FIXEDROT;
	if (def->allow_sleep) b.flags |= BF_AUTOSLEEP;
	if (def->awake) b.flags |= BF_AWAKE;
	if (def->active) b.flags |= BF_ACTIVE;
	b.px = def->px;
	b.py = def->py;
	b.qs = sinf(def->angle); // b2Rot::Set (b2Math.h:294-299)
	b.qc = cosf(def->angle);
	b.lcx = b.lcy = 0.0f;
	b.c0x = b.cx = def->px;
	b.c0y = b.cy = def->py;
	b.a0 = b.a = def->angle;
	b.vx = def->vx;
	b.vy = def->vy;
	b.w = def->w;
	b.linearDamping = def->linear_damping;
	b.angularDamping = def->angular_damping;
	b.gravityScale = def->gravity_scale;
	b.fx = b.fy = b.torque = 0.0f;
	b.sleepTime = 0.0f;
	b.pullEpoch = w->mirrorEpoch;
	b.forceEpoch = w->stepEpoch;
	if (def->type == B2HIP_DYNAMIC_BODY)
	{
		b.mass = 1.0f;
		b.invMass = 1.0f;
	}
	else
	{
		b.mass = 0.0f;
		b.invMass = 0.0f;
	}
	b.I = 0.0f;
	b.invI = 0.0f;
	b.dirty = true;
	b.worldIndex = -1;
	if (def->type != B2HIP_STATIC_BODY)
	{
		// b2World::CreateBody (b2World.cpp:571-575): appended to m_nonStaticBodies
		b.worldIndex = (int)w->nonStatic.size();
		w->nonStatic.push_back((int)w->bodies.size());
		w->orderDirty = true;
	}
	w->bodies.push_back(b);
	if (w->spatial) w->spOwnersDirty = true; // (the new body falls into the strip of its x at the next step)
	w->dirtyList.push_back((int)w->bodies.size() - 1);
	return (int)w->bodies.size() - 1;
}

int b2hip_create_fixture(b2hip_world* w, int body, const b2hip_fixture_def* def, const b2hip_shape* shape)
{
	if (!w || !def || !shape) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rcu = checkUsable(w, "b2hip_create_fixture", true)) return rcu;
	if (body < 0 || body >= (int)w->bodies.size()) return setError(B2HIP_ERR_INVALID, "bad body id");
	if (shape->type != B2HIP_SHAPE_CIRCLE && shape->type != B2HIP_SHAPE_EDGE && shape->type != B2HIP_SHAPE_POLYGON && shape->type != B2HIP_SHAPE_CHAIN)
	{
		return setError(B2HIP_ERR_INVALID, "unknown shape type");
	}
	ShapeRec rec;
	memset(&rec, 0, sizeof(rec));
	rec.type = shape->type;
	rec.count = shape->count;
	rec.radius = shape->radius;
	rec.centroid = v2(shape->centroid[0], shape->centroid[1]);
	int nv = shape->type == B2HIP_SHAPE_POLYGON ? shape->count : (B2D_IS_SEGMENT(shape->type) ? 4 : 1);
	if (nv > B2D_MAX_POLY_VERTS) return setError(B2HIP_ERR_INVALID, "too many polygon vertices");
	for (int i = 0; i < nv; ++i)
	{
		rec.verts[i] = v2(shape->verts[2 * i], shape->verts[2 * i + 1]);
		if (shape->type == B2HIP_SHAPE_POLYGON) rec.normals[i] = v2(shape->normals[2 * i], shape->normals[2 * i + 1]);
	}
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	HostFixture f;
	memset(&f, 0, sizeof(f));
	f.body = body;
	f.shape = internShape(w, rec);
	f.density = def->density;
	f.friction = def->friction;
	f.restitution = def->restitution;
	f.categoryBits = def->category_bits;
	f.maskBits = def->mask_bits;
	f.groupIndex = def->group_index;
	f.isSensor = def->is_sensor != 0;
	f.thick = def->thick_shape != 0;
	// b2Fixture::CreateProxies (b2Fixture.cpp:126-141) + b2DynamicTree::CreateProxy (b2DynamicTree.cpp:105-119)
	AABB aabb = b2dShapeAABB(&rec, hostXf(b));
	f.fat[0] = aabb.lo.x - B2D_AABB_EXTENSION;
	f.fat[1] = aabb.lo.y - B2D_AABB_EXTENSION;
	f.fat[2] = aabb.hi.x + B2D_AABB_EXTENSION;
	f.fat[3] = aabb.hi.y + B2D_AABB_EXTENSION;
	const bool bodyActive = (b.flags & BF_ACTIVE) != 0;
	if (bodyActive)
	{
		f.proxyKey = allocProxyKey(w);
		if (f.proxyKey < 0) return setError(B2HIP_ERR_UNSUPPORTED, "proxy id reuse after the broad-phase tree was emptied is not modelled");
	}
	else
	{
		// (b2Body.cpp:199-203: an inactive body's fixtures get their proxies when it is activated)
		f.proxyKey = -1;
		f.noProxy = true;
	}
	const int id = (int)w->fixtures.size();
	w->fixtures.push_back(f);
	b.fixtures.push_back(id);
	if (bodyActive) w->pendingMoves.push_back(id);
	if (f.density > 0.0f)
	{
		resetMassData(w, b);
	}
	w->newFixture = true;
	return id;
}

int b2hip_create_revolute_joint(b2hip_world* w, const b2hip_revolute_joint_def* def)
{
	if (!w || !def) return setError(B2HIP_ERR_INVALID, "null argument");
	const int nb = (int)w->bodies.size();
	if (def->body_a < 0 || def->body_a >= nb || def->body_b < 0 || def->body_b >= nb) return setError(B2HIP_ERR_INVALID, "bad body id");
	RevoluteJoint j;
	memset(&j, 0, sizeof(j));
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.referenceAngle = def->reference_angle;
	j.enableLimit = def->enable_limit;
	j.lowerAngle = def->lower_angle;
	j.upperAngle = def->upper_angle;
	j.enableMotor = def->enable_motor;
	j.motorSpeed = def->motor_speed;
	j.maxMotorTorque = def->max_motor_torque;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_distance_joint(b2hip_world* w, const b2hip_distance_joint_def* def)
{
	if (!w || !def) return setError(B2HIP_ERR_INVALID, "null argument");
	const int nb = (int)w->bodies.size();
	if (def->body_a < 0 || def->body_a >= nb || def->body_b < 0 || def->body_b >= nb) return setError(B2HIP_ERR_INVALID, "bad body id");
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_DISTANCE;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.length = def->length;
	j.frequencyHz = def->frequency_hz;
	j.dampingRatio = def->damping_ratio;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_prismatic_joint(b2hip_world* w, const b2hip_prismatic_joint_def* def)
{
	if (!w || !def) return setError(B2HIP_ERR_INVALID, "null argument");
	const int nb = (int)w->bodies.size();
	if (def->body_a < 0 || def->body_a >= nb || def->body_b < 0 || def->body_b >= nb) return setError(B2HIP_ERR_INVALID, "bad body id");
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_PRISMATIC;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.localAxisA = v2(def->local_axis_a[0], def->local_axis_a[1]);
	b2dNormalize(j.localAxisA); // b2PrismaticJoint.cpp:104
	j.referenceAngle = def->reference_angle;
	j.enableLimit = def->enable_limit;
	j.lowerTranslation = def->lower_translation;
	j.upperTranslation = def->upper_translation;
	j.enableMotor = def->enable_motor;
	j.motorSpeed = def->motor_speed;
	j.maxMotorForce = def->max_motor_force;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_weld_joint(b2hip_world* w, const b2hip_weld_joint_def* def)
{
	if (!w || !def) return setError(B2HIP_ERR_INVALID, "null argument");
	const int nb = (int)w->bodies.size();
	if (def->body_a < 0 || def->body_a >= nb || def->body_b < 0 || def->body_b >= nb) return setError(B2HIP_ERR_INVALID, "bad body id");
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_WELD;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.referenceAngle = def->reference_angle;
	j.frequencyHz = def->frequency_hz;
	j.dampingRatio = def->damping_ratio;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

static int checkJointBodies(b2hip_world* w, int a, int b)
{
	const int nb = w ? (int)w->bodies.size() : 0;
	if (!w) return setError(B2HIP_ERR_INVALID, "null argument");
	if (a < 0 || a >= nb || b < 0 || b >= nb) return setError(B2HIP_ERR_INVALID, "bad body id");
	return 0;
}

int b2hip_create_wheel_joint(b2hip_world* w, const b2hip_wheel_joint_def* def)
{
	if (!def) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rc = checkJointBodies(w, def->body_a, def->body_b)) return rc;
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_WHEEL;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.localAxisA = v2(def->local_axis_a[0], def->local_axis_a[1]);
	j.frequencyHz = def->frequency_hz;
	j.dampingRatio = def->damping_ratio;
	j.enableMotor = def->enable_motor;
	j.motorSpeed = def->motor_speed;
	j.maxMotorTorque = def->max_motor_torque;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_rope_joint(b2hip_world* w, const b2hip_rope_joint_def* def)
{
	if (!def) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rc = checkJointBodies(w, def->body_a, def->body_b)) return rc;
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_ROPE;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.maxLength = def->max_length;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_friction_joint(b2hip_world* w, const b2hip_friction_joint_def* def)
{
	if (!def) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rc = checkJointBodies(w, def->body_a, def->body_b)) return rc;
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_FRICTION;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.maxForce = def->max_force;
	j.maxTorque = def->max_torque;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_motor_joint(b2hip_world* w, const b2hip_motor_joint_def* def)
{
	if (!def) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rc = checkJointBodies(w, def->body_a, def->body_b)) return rc;
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_MOTOR;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.linearOffset = v2(def->linear_offset[0], def->linear_offset[1]);
	j.angularOffset = def->angular_offset;
	j.maxForce = def->max_force;
	j.maxTorque = def->max_torque;
	j.correctionFactor = def->correction_factor;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_pulley_joint(b2hip_world* w, const b2hip_pulley_joint_def* def)
{
	if (!def) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rc = checkJointBodies(w, def->body_a, def->body_b)) return rc;
	if (def->ratio == 0.0f) return setError(B2HIP_ERR_INVALID, "pulley ratio must not be zero");
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_PULLEY;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.groundAnchorA = v2(def->ground_anchor_a[0], def->ground_anchor_a[1]);
	j.s1 = def->ground_anchor_b[0];
	j.s2 = def->ground_anchor_b[1];
	j.ratio = def->ratio;
	j.constant = def->length_a + def->ratio * def->length_b; // b2PulleyJoint.cpp:75
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_mouse_joint(b2hip_world* w, const b2hip_mouse_joint_def* def)
{
	if (!def) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rc = checkJointBodies(w, def->body_a, def->body_b)) return rc;
	HostBody& bB = w->bodies[def->body_b];
	if (!bB.dirty) pullBody(w, def->body_b); // current transform of bodyB
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_MOUSE;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.targetA = v2(def->target[0], def->target[1]);
	// m_localAnchorB = b2MulT(bodyB transform, target) (b2MouseJoint.cpp:45)
	const float px = def->target[0] - bB.px, py = def->target[1] - bB.py;
	j.localAnchorB = v2(bB.qc * px + bB.qs * py, -bB.qs * px + bB.qc * py);
	j.bodyMass = bB.mass;
	j.maxForce = def->max_force;
	j.frequencyHz = def->frequency_hz;
	j.dampingRatio = def->damping_ratio;
	j.collideConnected = def->collide_connected;
	w->nMouseJoints += 1;
	return addJoint(w, j);
}

int b2hip_joint_set_target(b2hip_world* w, int joint, float x, float y)
{
	if (int rcu = checkUsable(w, "b2hip_joint_set_target", true)) return rcu;
	if (!w || joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	JointRec& j = w->joints[joint];
	if (j.type != B2D_JOINT_MOUSE) return setError(B2HIP_ERR_INVALID, "not a mouse joint");
	if (x == j.targetA.x && y == j.targetA.y) return 0;
	setAwake(w, j.bodyB);
	j.targetA = v2(x, y);
	w->jointEdits.push_back(std::make_pair(joint, 2));
	return 0;
}

// b2GearJoint::b2GearJoint (b2GearJoint.cpp:50-129): everything is derived from the two joints and the bodies' current poses
static float gearCoordinate(b2hip_world* w, const JointRec& jt, int moving, int fixed)
{
	const HostBody& bm = w->bodies[moving];
	const HostBody& bf = w->bodies[fixed];
	if (jt.type == B2D_JOINT_REVOLUTE) return bm.a - bf.a - jt.referenceAngle;
	// pA = b2MulT(xfC.q, b2Mul(xfA.q, m_localAnchorA) + (xfA.p - xfC.p)); coordinate = b2Dot(pA - pC, m_localAxisC)
	const V2 la = jt.localAnchorB, lc = jt.localAnchorA;
	const V2 wa = v2(bm.qc * la.x - bm.qs * la.y, bm.qs * la.x + bm.qc * la.y) + v2(bm.px - bf.px, bm.py - bf.py);
	const V2 pa = v2(bf.qc * wa.x + bf.qs * wa.y, -bf.qs * wa.x + bf.qc * wa.y);
	return b2dDot(pa - lc, jt.localAxisA);
}

int b2hip_create_gear_joint(b2hip_world* w, const b2hip_gear_joint_def* def)
{
	if (!w || !def) return setError(B2HIP_ERR_INVALID, "null argument");
	const int nj = (int)w->joints.size();
	if (def->joint1 < 0 || def->joint1 >= nj || def->joint2 < 0 || def->joint2 >= nj) return setError(B2HIP_ERR_INVALID, "bad joint id");
	const JointRec j1 = w->joints[def->joint1], j2 = w->joints[def->joint2];
	if ((j1.type != B2D_JOINT_REVOLUTE && j1.type != B2D_JOINT_PRISMATIC) || (j2.type != B2D_JOINT_REVOLUTE && j2.type != B2D_JOINT_PRISMATIC))
		return setError(B2HIP_ERR_INVALID, "a gear joint connects revolute and / or prismatic joints");
	const int ids[4] = { j1.bodyB, j2.bodyB, j1.bodyA, j2.bodyA }; // A, B, C, D
	for (int k = 0; k < 4; ++k)
		if (!w->bodies[ids[k]].dirty) pullBody(w, ids[k]);
	GearRec g;
	memset(&g, 0, sizeof(g));
	g.bodyC = ids[2];
	g.bodyD = ids[3];
	g.typeA = j1.type;
	g.typeB = j2.type;
	g.localAnchorC = j1.localAnchorA; g.localAnchorA = j1.localAnchorB; g.referenceAngleA = j1.referenceAngle;
	g.localAxisC = j1.type == B2D_JOINT_PRISMATIC ? j1.localAxisA : v2(0.0f, 0.0f);
	g.localAnchorD = j2.localAnchorA; g.localAnchorB = j2.localAnchorB; g.referenceAngleB = j2.referenceAngle;
	g.localAxisD = j2.type == B2D_JOINT_PRISMATIC ? j2.localAxisA : v2(0.0f, 0.0f);
	const float coordinateA = gearCoordinate(w, j1, ids[0], ids[2]);
	const float coordinateB = gearCoordinate(w, j2, ids[1], ids[3]);
	g.ratio = def->ratio;
	g.constant = coordinateA + g.ratio * coordinateB;
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_GEAR;
	j.bodyA = ids[0];
	j.bodyB = ids[1];
	j.enableLimit = (int)w->gears.size();
	j.collideConnected = def->collide_connected;
	w->gears.push_back(g);
	return addJoint(w, j);
}

// b2Body::SetAwake(true) on both bodies of a joint whose definition changed (b2RevoluteJoint.cpp:418-500)
static void wakeJointBodies(b2hip_world* w, const JointRec& j)
{
	setAwake(w, j.bodyA);
	setAwake(w, j.bodyB);
}

int b2hip_destroy_joint(b2hip_world* w, int joint)
{
	if (!w || joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	if (int rcu = checkUsable(w, "b2hip_destroy_joint", true)) return rcu;
	JointRec& j = w->joints[joint];
	if (j.type == B2D_JOINT_DEAD) return setError(B2HIP_ERR_INVALID, "joint already destroyed");
	// (a gear joint must be destroyed before the joints it couples, as in the reference)
	wakeJointBodies(w, j);
	// contacts between the two bodies are filtered again when the joint kept them from colliding (b2World.cpp:833-845)
	if (j.collideConnected == 0) w->pendingFilter.push_back(std::make_pair(j.bodyA, j.bodyB));
	if (j.type == B2D_JOINT_MOUSE) w->nMouseJoints -= 1;
	j.type = B2D_JOINT_DEAD;
	w->jadjJoints = (size_t)-1; // per-body joint lists are rebuilt without it
	w->jointEdits.push_back(std::make_pair(joint, 3));
	return 0;
}

int b2hip_joint_set_motor(b2hip_world* w, int joint, int enable_motor, float motor_speed, float max_motor)
{
	if (int rcu = checkUsable(w, "b2hip_joint_set_motor", true)) return rcu;
	if (!w || joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	JointRec& j = w->joints[joint];
	if (j.type != B2D_JOINT_REVOLUTE && j.type != B2D_JOINT_PRISMATIC && j.type != B2D_JOINT_WHEEL)
		return setError(B2HIP_ERR_INVALID, "joint type has no motor");
	if ((enable_motor != 0) == (j.enableMotor != 0) && motor_speed == j.motorSpeed && max_motor == j.maxMotorTorque) return 0;
	wakeJointBodies(w, j);
	j.enableMotor = enable_motor != 0;
	j.motorSpeed = motor_speed;
	j.maxMotorTorque = max_motor;
	w->jointEdits.push_back(std::make_pair(joint, 0));
	return 0;
}

int b2hip_joint_set_offsets(b2hip_world* w, int joint, float linear_x, float linear_y, float angular)
{
	if (int rcu = checkUsable(w, "b2hip_joint_set_offsets", true)) return rcu;
	if (!w || joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	JointRec& j = w->joints[joint];
	if (j.type != B2D_JOINT_MOTOR) return setError(B2HIP_ERR_INVALID, "not a motor joint");
	if (linear_x == j.linearOffset.x && linear_y == j.linearOffset.y && angular == j.angularOffset) return 0;
	wakeJointBodies(w, j);
	j.linearOffset = v2(linear_x, linear_y);
	j.angularOffset = angular;
	w->jointEdits.push_back(std::make_pair(joint, 2)); // rewrite the anchor / offset members
	return 0;
}

int b2hip_joint_set_limits(b2hip_world* w, int joint, int enable_limit, float lower, float upper)
{
	if (int rcu = checkUsable(w, "b2hip_joint_set_limits", true)) return rcu;
	if (!w || joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	JointRec& j = w->joints[joint];
	if (j.type != B2D_JOINT_REVOLUTE && j.type != B2D_JOINT_PRISMATIC) return setError(B2HIP_ERR_INVALID, "joint type has no limits");
	if (lower > upper) return setError(B2HIP_ERR_INVALID, "lower limit above upper limit");
	if ((enable_limit != 0) == (j.enableLimit != 0) && lower == j.lowerAngle && upper == j.upperAngle) return 0;
	wakeJointBodies(w, j);
	j.enableLimit = enable_limit != 0;
	j.lowerAngle = lower;
	j.upperAngle = upper;
	w->jointEdits.push_back(std::make_pair(joint, 1)); // the limit impulse restarts from zero
	return 0;
}

// b2RopeJoint::GetLimitState (b2RopeJoint.h:84) and the limit state of revolute / prismatic joints: the solver's, from the
// device record; 0 inactive, 1 at lower, 2 at upper, 3 equal (b2LimitState, b2Joint.h:58-64); negative: error
int b2hip_get_joint_limit_state(b2hip_world* w, int joint)
{
	if (int rcu = checkUsable(w, "b2hip_get_joint_limit_state", true)) return rcu;
	if (joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	DEVICE_GUARD(w);
	int state = w->joints[joint].limitState;
	if ((size_t)joint < w->upJoints && w->d_joints.p != nullptr)
	{
		HIP_TRY(hipMemcpyAsync(&state, (const char*)(w->d_joints.p + joint) + offsetof(JointRec, limitState), sizeof(int), hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
	}
	return state;
}

int b2hip_get_joint_reaction(b2hip_world* w, int joint, float inv_dt, float out4[4])
{
	if (int rcu = checkUsable(w, "b2hip_get_joint_reaction", true)) return rcu;
	if (!w || !out4 || joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	DEVICE_GUARD(w);
	JointRec rec = w->joints[joint];
	GearRec gear;
	memset(&gear, 0, sizeof(gear));
	const bool isGear = rec.type == B2D_JOINT_GEAR;
	// (the solver's state lives in the device copy; a joint the device has not seen yet has done nothing)
	if ((size_t)joint < w->upJoints && w->d_joints.p != nullptr)
	{
		HIP_TRY(hipMemcpyAsync(&rec, w->d_joints.p + joint, sizeof(JointRec), hipMemcpyDeviceToHost, w->stream));
		if (isGear && (size_t)rec.enableLimit < w->upGears) HIP_TRY(hipMemcpyAsync(&gear, w->d_gears.p + rec.enableLimit, sizeof(GearRec), hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
	}
	const JointReaction r = b2dJointReaction(&rec, isGear ? &gear : nullptr, inv_dt);
	out4[0] = r.force.x;
	out4[1] = r.force.y;
	out4[2] = r.torque;
	out4[3] = r.motor;
	return 0;
}

int b2hip_body_count(const b2hip_world* w)
{
	return w ? (int)w->bodies.size() : 0;
}

int b2hip_fixture_count(const b2hip_world* w)
{
	return w ? (int)w->fixtures.size() : 0;
}

int b2hip_get_mass_data(const b2hip_world* w, int body, b2hip_mass_data* out)
{
	if (!w || !out || body < 0 || body >= (int)w->bodies.size()) return setError(B2HIP_ERR_INVALID, "bad argument");
	const HostBody& b = w->bodies[body];
	out->mass = b.mass;
	// b2Body::GetInertia (b2Body.h:585-588)
	out->inertia = b.I + b.mass * (b.lcx * b.lcx + b.lcy * b.lcy);
	out->local_center[0] = b.lcx;
	out->local_center[1] = b.lcy;
	out->inv_mass = b.invMass;
	out->inv_inertia = b.invI;
	return 0;
}

int b2hip_apply_force(b2hip_world* w, int body, float fx, float fy, float torque, int wake)
{
	if (int rcu = checkUsable(w, "b2hip_apply_force", true)) return rcu;
	if (!w || body < 0 || body >= (int)w->bodies.size()) return setError(B2HIP_ERR_INVALID, "bad argument");
	if (w->bodies[body].type != B2HIP_DYNAMIC_BODY) return 0;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	// (a row that was already dirty - an edit from a callback of the last step - has not been through pullBody: the forces it
	// holds are the last step's, which the step cleared on the device, and its epoch must say that THIS force is new; else a
	// later pull of the same step - a PreSolve edit - takes the force for a stale one and drops it)
	if (w->def.auto_clear_forces && b.forceEpoch != w->stepEpoch) { b.fx = b.fy = b.torque = 0.0f; }
	b.forceEpoch = w->stepEpoch;
	if (wake && (b.flags & BF_AWAKE) == 0)
	{
		b.flags |= BF_AWAKE;
		b.sleepTime = 0.0f;
	}
	if (b.flags & BF_AWAKE)
	{
		b.fx += fx;
		b.fy += fy;
		b.torque += torque;
	}
	return 0;
}

int b2hip_set_velocity(b2hip_world* w, int body, float vx, float vy, float omega)
{
	if (int rcu = checkUsable(w, "b2hip_set_velocity", true)) return rcu;
	if (!w || body < 0 || body >= (int)w->bodies.size()) return setError(B2HIP_ERR_INVALID, "bad argument");
	if (w->bodies[body].type == B2HIP_STATIC_BODY) return 0;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (vx * vx + vy * vy > 0.0f || omega * omega > 0.0f)
	{
		b.flags |= BF_AWAKE;
		b.sleepTime = 0.0f;
	}
	b.vx = vx;
	b.vy = vy;
	b.w = omega;
	return 0;
}

static int stepBeginImpl(b2hip_world* w, float dt, int velocity_iterations, int position_iterations)
{
	int rc = flushEdits(w);
	if (rc) return rc;
	StepParams& sp = w->sp;
	sp.dt = dt;
	sp.inv_dt = dt > 0.0f ? 1.0f / dt : 0.0f;
	sp.dtRatio = w->inv_dt0 * dt;
	sp.velIters = velocity_iterations;
	sp.posIters = position_iterations;
	sp.warmStarting = w->def.warm_starting;
	sp.allowSleep = w->def.allow_sleep;
	sp.gravity = v2(w->def.gravity_x, w->def.gravity_y);
	if (w->kernelTiming > 1)
	{
		w->ktUsed = 0;
		w->ktKind = 0;
	}
	w->stepActive = true;
	w->stepSolves = w->stepComplete;
	w->dw.toiContinue = w->stepComplete ? 0 : 1;
	w->dw.toiEventCap = w->def.sub_stepping ? 1 : 0;
	// zero the per-step counters (keep nContacts / nMoves / cur)
	Counters zero;
	memset(&zero, 0, sizeof(zero));
	// (b2Profile::step: from the start of this kernel to the end of k_end_step, slots 14 and 13)
	w->dw.stampMask = 1u << 14;
	LAUNCH(w, k_step_begin, 1, 64, w->dw, w->gridBar.p);
	w->restArrived = 0; // (bar[5] starts the step at 0)
	w->toiCountersFresh = true;
	rc = applyPendingFilters(w);
	if (rc) return rc;
	rc = applyEditOps(w, false); // (after k_step_begin: the end events of destroyed contacts belong to this step's list)
	if (rc) return rc;
	if (w->spatial)
	{
		if (listenerOn(w) || hasFilter(w) || w->def.sub_stepping) return setError(B2HIP_ERR_UNSUPPORTED, "contact listeners, filters and sub-stepping are not supported in a spatially sharded world");
		rc = spBeginStep(w);
		if (rc) return rc;
	}
	stampPhase(w, 0);
	// b2World.cpp:1628-1639: new fixtures -> find their contacts before colliding
	if (w->newFixture)
	{
		rc = findNewContacts(w, true);
		if (rc) return rc;
		w->newFixture = false;
	}
	stampPhase(w, 1);
	return 0;
}

// ---- life cycle and mutators ---------------------------------------------------------------------------------------------
static int checkBody(b2hip_world* w, int body, const char* what)
{
	if (int rcu = checkUsable(w, what, true)) return rcu;
	if (body < 0 || body >= (int)w->bodies.size() || w->bodies[body].dead) return setError(B2HIP_ERR_INVALID, std::string(what) + ": bad body id");
	return 0;
}

static int checkFixture(b2hip_world* w, int fixture, const char* what)
{
	if (int rcu = checkUsable(w, what, true)) return rcu;
	if (fixture < 0 || fixture >= (int)w->fixtures.size() || w->fixtures[fixture].dead) return setError(B2HIP_ERR_INVALID, std::string(what) + ": bad fixture id");
	return 0;
}

static void queueOp(b2hip_world* w, int kind, int id)
{
	w->editOps.push_back(make_int2(kind, id));
}

// b2Fixture::DestroyProxies (b2Fixture.cpp:143-157) + the host bookkeeping of a fixture that is gone
static void dropFixture(b2hip_world* w, int fixture)
{
	HostFixture& f = w->fixtures[fixture];
	freeProxyKey(w, f.proxyKey);
	f.dead = true;
	w->proxyEdits.push_back(fixture);
	w->proxyListsStale = true;
	// (a proxy created since the last step and not yet buffered on the device leaves the pending moves too: UnBufferMove)
	w->pendingMoves.erase(std::remove(w->pendingMoves.begin(), w->pendingMoves.end(), fixture), w->pendingMoves.end());
}

int b2hip_destroy_fixture(b2hip_world* w, int fixture)
{
	if (int rc = checkFixture(w, fixture, "b2hip_destroy_fixture")) return rc;
	const int body = w->fixtures[fixture].body;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	queueOp(w, EDIT_DESTROY_FIXTURE, fixture);
	b.fixtures.erase(std::remove(b.fixtures.begin(), b.fixtures.end(), fixture), b.fixtures.end());
	dropFixture(w, fixture);
	resetMassData(w, b);
	return B2HIP_OK;
}

int b2hip_destroy_body(b2hip_world* w, int body)
{
	if (int rc = checkBody(w, body, "b2hip_destroy_body")) return rc;
	// joints first, newest first (the body's joint list is newest first, b2World.cpp:697-710)
	for (int j = (int)w->joints.size() - 1; j >= 0; --j)
	{
		if (w->joints[j].type == B2D_JOINT_DEAD) continue;
		bool touches = w->joints[j].bodyA == body || w->joints[j].bodyB == body;
		if (w->joints[j].type == B2D_JOINT_GEAR)
		{
			const GearRec& g = w->gears[w->joints[j].enableLimit];
			touches = touches || g.bodyC == body || g.bodyD == body;
		}
		if (touches)
		{
			const int rc = b2hip_destroy_joint(w, j);
			if (rc) return rc;
		}
	}
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	queueOp(w, EDIT_DESTROY_BODY, body);
	for (int k = (int)b.fixtures.size() - 1; k >= 0; --k) dropFixture(w, b.fixtures[k]); // newest first
	b.fixtures.clear();
	if (b.worldIndex >= 0)
	{
		// b2RemoveAndSwapBack on m_nonStaticBodies (b2World.cpp:662-667)
		const int slot = b.worldIndex, last = w->nonStatic.back();
		w->nonStatic[(size_t)slot] = last;
		w->bodies[last].worldIndex = slot;
		w->nonStatic.pop_back();
		b.worldIndex = -1;
		w->orderDirty = true;
	}
	b.dead = 1;
	b.type = B2HIP_STATIC_BODY;
	b.flags &= ~(BF_ACTIVE | BF_AWAKE | BF_BULLET);
	b.vx = b.vy = b.w = 0.0f;
	b.fx = b.fy = b.torque = 0.0f;
	b.invMass = b.invI = 0.0f;
	return B2HIP_OK;
}

int b2hip_body_is_destroyed(const b2hip_world* w, int body)
{
	return w && body >= 0 && body < (int)w->bodies.size() && w->bodies[body].dead ? 1 : 0;
}

int b2hip_fixture_is_destroyed(const b2hip_world* w, int fixture)
{
	return w && fixture >= 0 && fixture < (int)w->fixtures.size() && w->fixtures[fixture].dead ? 1 : 0;
}

// The fat AABB a fixture's proxy has right now (the device owns it once the fixture is uploaded)
static int currentFat(b2hip_world* w, int fixture, float out4[4])
{
	if ((size_t)fixture >= w->upFixtures || std::find(w->fatEdits.begin(), w->fatEdits.end(), fixture) != w->fatEdits.end())
	{
		memcpy(out4, w->fixtures[fixture].fat, 16);
		return 0;
	}
	DEVICE_GUARD(w);
	HIP_TRY(hipStreamSynchronize(w->stream));
	HIP_TRY(hipMemcpy(out4, w->p_fat.p + fixture, 16, hipMemcpyDeviceToHost));
	return 0;
}

int b2hip_set_transform(b2hip_world* w, int body, float x, float y, float angle)
{
	if (int rc = checkBody(w, body, "b2hip_set_transform")) return rc;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	b.qs = sinf(angle);
	b.qc = cosf(angle);
	b.px = x;
	b.py = y;
	const V2 c = b2dMulXV(hostXf(b), v2(b.lcx, b.lcy));
	b.cx = b.c0x = c.x;
	b.cy = b.c0y = c.y;
	b.a = b.a0 = angle;
	b.resetSweep = 1;
	// b2Fixture::Synchronize(broadPhase, xf, xf) for every fixture, newest first -> b2DynamicTree::MoveProxy with zero displacement
	for (int k = (int)b.fixtures.size() - 1; k >= 0; --k)
	{
		const int id = b.fixtures[k];
		HostFixture& f = w->fixtures[id];
		float fat[4];
		if (int rc = currentFat(w, id, fat)) return rc;
		const AABB aabb = b2dShapeAABB(&w->shapes[f.shape], hostXf(b));
		if (fat[0] <= aabb.lo.x && fat[1] <= aabb.lo.y && aabb.hi.x <= fat[2] && aabb.hi.y <= fat[3])
		{
			memcpy(f.fat, fat, 16);
			continue;
		}
		f.fat[0] = aabb.lo.x - B2D_AABB_EXTENSION;
		f.fat[1] = aabb.lo.y - B2D_AABB_EXTENSION;
		f.fat[2] = aabb.hi.x + B2D_AABB_EXTENSION;
		f.fat[3] = aabb.hi.y + B2D_AABB_EXTENSION;
		w->proxyEdits.push_back(id);
		w->fatEdits.push_back(id);
		if ((size_t)id < w->upFixtures || std::find(w->pendingMoves.begin(), w->pendingMoves.end(), id) == w->pendingMoves.end()) w->pendingMoves.push_back(id);
		w->newFixture = w->newFixture; // (moves alone do not ask for the top-of-step pair update: the end-of-step one takes them)
	}
	return B2HIP_OK;
}

int b2hip_set_active(b2hip_world* w, int body, int active)
{
	if (int rc = checkBody(w, body, "b2hip_set_active")) return rc;
	if (((w->bodies[body].flags & BF_ACTIVE) != 0) == (active != 0)) return B2HIP_OK;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (active)
	{
		b.flags |= BF_ACTIVE;
		// b2Fixture::CreateProxies for every fixture, newest first: fat AABB at the body's transform, a fresh proxy id, a buffered move
		for (int k = (int)b.fixtures.size() - 1; k >= 0; --k)
		{
			const int id = b.fixtures[k];
			HostFixture& f = w->fixtures[id];
			const AABB aabb = b2dShapeAABB(&w->shapes[f.shape], hostXf(b));
			f.fat[0] = aabb.lo.x - B2D_AABB_EXTENSION;
			f.fat[1] = aabb.lo.y - B2D_AABB_EXTENSION;
			f.fat[2] = aabb.hi.x + B2D_AABB_EXTENSION;
			f.fat[3] = aabb.hi.y + B2D_AABB_EXTENSION;
			f.proxyKey = allocProxyKey(w);
			if (f.proxyKey < 0) return setError(B2HIP_ERR_UNSUPPORTED, "proxy id reuse after the broad-phase tree was emptied is not modelled");
			f.noProxy = false;
			w->proxyEdits.push_back(id);
			w->fatEdits.push_back(id);
			w->pendingMoves.push_back(id);
		}
		w->proxyListsStale = true;
		return B2HIP_OK;
	}
	b.flags &= ~BF_ACTIVE;
	// b2Fixture::DestroyProxies, newest fixture first, then the body's contacts in its contact-list order
	for (int k = (int)b.fixtures.size() - 1; k >= 0; --k)
	{
		const int id = b.fixtures[k];
		HostFixture& f = w->fixtures[id];
		if (f.noProxy) continue;
		freeProxyKey(w, f.proxyKey);
		f.proxyKey = -1;
		f.noProxy = true;
		w->proxyEdits.push_back(id);
		w->pendingMoves.erase(std::remove(w->pendingMoves.begin(), w->pendingMoves.end(), id), w->pendingMoves.end());
	}
	w->proxyListsStale = true;
	queueOp(w, EDIT_DESTROY_BODY, body);
	return B2HIP_OK;
}

int b2hip_set_type(b2hip_world* w, int body, int type)
{
	if (int rc = checkBody(w, body, "b2hip_set_type")) return rc;
	if (type < B2HIP_STATIC_BODY || type > B2HIP_DYNAMIC_BODY) return setError(B2HIP_ERR_INVALID, "b2hip_set_type: bad body type");
	if (w->bodies[body].type == type) return B2HIP_OK;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (b.type == B2HIP_STATIC_BODY)
	{
		// out of m_staticBodies, to the end of m_nonStaticBodies (b2Body.cpp:131-140)
		b.worldIndex = (int)w->nonStatic.size();
		w->nonStatic.push_back(body);
		w->orderDirty = true;
	}
	b.type = type;
	resetMassData(w, b);
	b.resetSweep = 1; // (the mass data moved the sweep origin with the centre)
	if (type == B2HIP_STATIC_BODY)
	{
		b.vx = b.vy = b.w = 0.0f;
		b.a0 = b.a;
		b.c0x = b.cx;
		b.c0y = b.cy;
		b.resetSweep = 1;
		// b2Body::SynchronizeFixtures with xf1 == xf (the sweep origin was just reset): MoveProxy with zero displacement
		for (int k = (int)b.fixtures.size() - 1; k >= 0; --k)
		{
			const int id = b.fixtures[k];
			HostFixture& f = w->fixtures[id];
			if (f.noProxy) continue;
			float fat[4];
			if (int rc = currentFat(w, id, fat)) return rc;
			const AABB aabb = b2dShapeAABB(&w->shapes[f.shape], hostXf(b));
			if (fat[0] <= aabb.lo.x && fat[1] <= aabb.lo.y && aabb.hi.x <= fat[2] && aabb.hi.y <= fat[3])
			{
				memcpy(f.fat, fat, 16);
				continue;
			}
			f.fat[0] = aabb.lo.x - B2D_AABB_EXTENSION;
			f.fat[1] = aabb.lo.y - B2D_AABB_EXTENSION;
			f.fat[2] = aabb.hi.x + B2D_AABB_EXTENSION;
			f.fat[3] = aabb.hi.y + B2D_AABB_EXTENSION;
			w->proxyEdits.push_back(id);
			w->fatEdits.push_back(id);
			w->pendingMoves.push_back(id);
		}
		// b2RemoveAndSwapBack on m_nonStaticBodies (b2Body.cpp:154-160)
		const int slot = b.worldIndex, last = w->nonStatic.back();
		w->nonStatic[(size_t)slot] = last;
		w->bodies[(size_t)last].worldIndex = slot;
		w->nonStatic.pop_back();
		b.worldIndex = -1;
		w->orderDirty = true;
	}
	b.flags |= BF_AWAKE;
	b.sleepTime = 0.0f;
	b.fx = b.fy = b.torque = 0.0f;
	// every contact of the body goes, in its contact-list order; TouchProxy on every proxy, newest fixture first
	queueOp(w, EDIT_DESTROY_BODY, body);
	for (int k = (int)b.fixtures.size() - 1; k >= 0; --k)
	{
		const int id = b.fixtures[k];
		if (!w->fixtures[id].noProxy) w->pendingMoves.push_back(id);
	}
	return B2HIP_OK;
}

int b2hip_set_awake(b2hip_world* w, int body, int awake)
{
	if (int rc = checkBody(w, body, "b2hip_set_awake")) return rc;
	if (awake)
	{
		setAwake(w, body);
		return B2HIP_OK;
	}
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	b.flags &= ~BF_AWAKE;
	b.sleepTime = 0.0f;
	b.vx = b.vy = b.w = 0.0f;
	b.fx = b.fy = b.torque = 0.0f;
	return B2HIP_OK;
}

int b2hip_set_bullet(b2hip_world* w, int body, int bullet)
{
	if (int rc = checkBody(w, body, "b2hip_set_bullet")) return rc;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	const bool was = (b.flags & BF_BULLET) != 0;
	if (bullet) b.flags |= BF_BULLET; else b.flags &= ~BF_BULLET;
	if (was != (bullet != 0)) queueOp(w, EDIT_RECALC_BODY, body);
	return B2HIP_OK;
}

int b2hip_apply_linear_impulse(b2hip_world* w, int body, float ix, float iy, float px, float py, int wake)
{
	if (int rc = checkBody(w, body, "b2hip_apply_linear_impulse")) return rc;
	if (w->bodies[body].type != B2HIP_DYNAMIC_BODY) return B2HIP_OK;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (wake && (b.flags & BF_AWAKE) == 0)
	{
		b.flags |= BF_AWAKE;
		b.sleepTime = 0.0f;
	}
	if (b.flags & BF_AWAKE)
	{
		// b2Body.h:915-921
		const float sx = b.invMass * ix, sy = b.invMass * iy;
		b.vx += sx;
		b.vy += sy;
		b.w += b.invI * ((px - b.cx) * iy - (py - b.cy) * ix);
	}
	return B2HIP_OK;
}

int b2hip_apply_linear_impulse_to_center(b2hip_world* w, int body, float ix, float iy, int wake)
{
	if (int rc = checkBody(w, body, "b2hip_apply_linear_impulse_to_center")) return rc;
	if (w->bodies[body].type != B2HIP_DYNAMIC_BODY) return B2HIP_OK;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (wake && (b.flags & BF_AWAKE) == 0)
	{
		b.flags |= BF_AWAKE;
		b.sleepTime = 0.0f;
	}
	if (b.flags & BF_AWAKE)
	{
		const float sx = b.invMass * ix, sy = b.invMass * iy; // b2Body.h:923-942
		b.vx += sx;
		b.vy += sy;
	}
	return B2HIP_OK;
}

int b2hip_apply_angular_impulse(b2hip_world* w, int body, float impulse, int wake)
{
	if (int rc = checkBody(w, body, "b2hip_apply_angular_impulse")) return rc;
	if (w->bodies[body].type != B2HIP_DYNAMIC_BODY) return B2HIP_OK;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (wake && (b.flags & BF_AWAKE) == 0)
	{
		b.flags |= BF_AWAKE;
		b.sleepTime = 0.0f;
	}
	if (b.flags & BF_AWAKE) b.w += b.invI * impulse;
	return B2HIP_OK;
}

int b2hip_fixture_set_sensor(b2hip_world* w, int fixture, int is_sensor)
{
	if (int rc = checkFixture(w, fixture, "b2hip_fixture_set_sensor")) return rc;
	HostFixture& f = w->fixtures[fixture];
	if (f.isSensor == (is_sensor != 0)) return B2HIP_OK;
	setAwake(w, f.body);
	f.isSensor = is_sensor != 0;
	w->proxyEdits.push_back(fixture);
	queueOp(w, EDIT_SENSOR_FIXTURE, fixture);
	queueOp(w, EDIT_RECALC_FIXTURE, fixture);
	return B2HIP_OK;
}

int b2hip_fixture_set_thick(b2hip_world* w, int fixture, int thick_shape)
{
	if (int rc = checkFixture(w, fixture, "b2hip_fixture_set_thick")) return rc;
	HostFixture& f = w->fixtures[fixture];
	if (f.thick == (thick_shape != 0)) return B2HIP_OK;
	f.thick = thick_shape != 0;
	w->proxyEdits.push_back(fixture);
	queueOp(w, EDIT_RECALC_FIXTURE, fixture);
	return B2HIP_OK;
}

int b2hip_fixture_refilter(b2hip_world* w, int fixture)
{
	if (int rc = checkFixture(w, fixture, "b2hip_fixture_refilter")) return rc;
	queueOp(w, EDIT_REFILTER_FIXTURE, fixture);
	w->refilterPending = true;
	// TouchProxy (b2BroadPhase.cpp:70-73): the proxy is buffered as moved so that new pairs can form
	if (w->bodies[w->fixtures[fixture].body].flags & BF_ACTIVE) w->pendingMoves.push_back(fixture);
	return B2HIP_OK;
}

int b2hip_fixture_set_filter(b2hip_world* w, int fixture, uint16_t category_bits, uint16_t mask_bits, int16_t group_index)
{
	if (int rc = checkFixture(w, fixture, "b2hip_fixture_set_filter")) return rc;
	HostFixture& f = w->fixtures[fixture];
	f.categoryBits = category_bits;
	f.maskBits = mask_bits;
	f.groupIndex = group_index;
	w->proxyEdits.push_back(fixture);
	return b2hip_fixture_refilter(w, fixture);
}

// b2Fixture::SetDensity / SetFriction / SetRestitution (b2Fixture.h:306-334): plain values - the density is read by the next
// ResetMassData, friction and restitution by the contacts created from now on (existing contacts keep their mixture)
int b2hip_fixture_set_material(b2hip_world* w, int fixture, float density, float friction, float restitution)
{
	if (int rc = checkFixture(w, fixture, "b2hip_fixture_set_material")) return rc;
	HostFixture& f = w->fixtures[fixture];
	f.density = density;
	f.friction = friction;
	f.restitution = restitution;
	w->proxyEdits.push_back(fixture);
	return B2HIP_OK;
}

// b2Body::SetLinearDamping / SetAngularDamping / SetGravityScale (b2Body.h:620-648): read by the next Solve
int b2hip_set_body_damping(b2hip_world* w, int body, float linear_damping, float angular_damping, float gravity_scale)
{
	if (int rc = checkBody(w, body, "b2hip_set_body_damping")) return rc;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	b.linearDamping = linear_damping;
	b.angularDamping = angular_damping;
	b.gravityScale = gravity_scale;
	return B2HIP_OK;
}

// b2Body::SetFixedRotation (b2Body.cpp:546-565): the flag, no spin, mass data again
int b2hip_set_fixed_rotation(b2hip_world* w, int body, int flag)
{
	if (int rc = checkBody(w, body, "b2hip_set_fixed_rotation")) return rc;
	HostBody& probe = w->bodies[body];
	if (((probe.flags & BF_FIXEDROT) != 0) == (flag != 0)) return B2HIP_OK;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (flag) b.flags |= BF_FIXEDROT; else b.flags &= ~BF_FIXEDROT;
	b.w = 0.0f;
	resetMassData(w, b);
	b.resetSweep = 1;
	return B2HIP_OK;
}

// b2Body::SetSleepingAllowed (b2Body.h:674-688): a body that may not sleep is woken
int b2hip_set_sleeping_allowed(b2hip_world* w, int body, int flag)
{
	if (int rc = checkBody(w, body, "b2hip_set_sleeping_allowed")) return rc;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (flag) b.flags |= BF_AUTOSLEEP;
	else
	{
		b.flags &= ~BF_AUTOSLEEP;
		b.flags |= BF_AWAKE; // SetAwake(true) (b2Body.h:690-718): the sleep timer restarts whether or not the body slept
		b.sleepTime = 0.0f;
	}
	return B2HIP_OK;
}

// b2Body::SetMassData (b2Body.cpp:387-424); mass_data == NULL: b2Body::ResetMassData (b2Body.cpp:310-385)
int b2hip_set_mass_data(b2hip_world* w, int body, const b2hip_mass_data* md)
{
	if (int rc = checkBody(w, body, "b2hip_set_mass_data")) return rc;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (md == nullptr)
	{
		resetMassData(w, b);
		b.resetSweep = 1;
		return B2HIP_OK;
	}
	if (b.type != B2HIP_DYNAMIC_BODY) return B2HIP_OK;
	b.invMass = 0.0f;
	b.I = 0.0f;
	b.invI = 0.0f;
	b.mass = md->mass;
	if (b.mass <= 0.0f) b.mass = 1.0f;
	b.invMass = 1.0f / b.mass;
	const V2 center = v2(md->local_center[0], md->local_center[1]);
	if (md->inertia > 0.0f && (b.flags & BF_FIXEDROT) == 0)
	{
		b.I = md->inertia - b.mass * b2dDot(center, center);
		b.invI = 1.0f / b.I;
	}
	const V2 oldCenter = v2(b.cx, b.cy);
	b.lcx = center.x;
	b.lcy = center.y;
	const V2 c = b2dMulXV(hostXf(b), center);
	b.c0x = b.cx = c.x;
	b.c0y = b.cy = c.y;
	const V2 dv = b2dCrossSV(b.w, c - oldCenter);
	b.vx += dv.x;
	b.vy += dv.y;
	b.resetSweep = 1;
	return B2HIP_OK;
}

// The scalar setters of the joint classes: plain assignments in the reference (b2DistanceJoint.h:117, b2RopeJoint.h:80,
// b2FrictionJoint.cpp:206-228, b2MotorJoint.cpp:222-251, b2MouseJoint.cpp:48-76, b2GearJoint.cpp:402-406)
int b2hip_joint_set_param(b2hip_world* w, int joint, int param, float value)
{
	if (int rcu = checkUsable(w, "b2hip_joint_set_param", true)) return rcu;
	if (joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	JointRec& j = w->joints[joint];
	const int t = j.type;
	int kind = 0;
	if (param == B2HIP_JOINT_LENGTH && (t == B2D_JOINT_DISTANCE || t == B2D_JOINT_ROPE)) { j.length = value; kind = 2; }
	else if (param == B2HIP_JOINT_MAX_FORCE && (t == B2D_JOINT_FRICTION || t == B2D_JOINT_MOTOR || t == B2D_JOINT_MOUSE)) j.maxForce = value;
	else if (param == B2HIP_JOINT_MAX_TORQUE && (t == B2D_JOINT_FRICTION || t == B2D_JOINT_MOTOR)) j.maxTorque = value;
	else if (param == B2HIP_JOINT_RATIO && t == B2D_JOINT_GEAR)
	{
		// (a gear's definition lives in its own record, GearRec; JointRec::enableLimit is its index there)
		const int gi = j.enableLimit;
		if (gi < 0 || gi >= (int)w->gears.size()) return setError(B2HIP_ERR_INVALID, "gear record missing");
		w->gears[(size_t)gi].ratio = value;
		if ((size_t)gi < w->upGears)
		{
			DEVICE_GUARD(w);
			HIP_TRY(hipMemcpy((char*)(w->d_gears.p + gi) + offsetof(GearRec, ratio), &value, sizeof(float), hipMemcpyHostToDevice));
		}
		return B2HIP_OK;
	}
	else if (param == B2HIP_JOINT_CORRECTION_FACTOR && t == B2D_JOINT_MOTOR) j.correctionFactor = value;
	else return setError(B2HIP_ERR_INVALID, "b2hip_joint_set_param: the joint's type has no such parameter");
	w->jointEdits.push_back(std::make_pair(joint, kind));
	return B2HIP_OK;
}

// b2World::ShiftOrigin (b2World.cpp:1862-1887)
int b2hip_shift_origin(b2hip_world* w, float x, float y)
{
	if (int rcu = checkUsable(w, "b2hip_shift_origin", true)) return rcu;
	if (w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_shift_origin inside a step");
	DEVICE_GUARD(w);
	// every edit made so far goes to the device first: from here on the device state is the one that is shifted
	int rc = flushEdits(w);
	if (rc) return rc;
	rc = applyEditOps(w, true);
	if (rc) return rc;
	const int n = std::max(std::max(w->dw.nBodies, w->dw.nProxies), std::max(w->dw.nJoints, 1));
	LAUNCH(w, k_shift_origin, gridFor(n), 256, w->dw, x, y);
	// the host's copies: joint records (uploaded again when a setter edits them), fat AABBs, and the body rows - read back
	for (size_t j = 0; j < w->joints.size(); ++j)
	{
		JointRec& jn = w->joints[j];
		if (jn.type == B2D_JOINT_MOUSE) { jn.targetA.x -= x; jn.targetA.y -= y; }
		else if (jn.type == B2D_JOINT_PULLEY) { jn.groundAnchorA.x -= x; jn.groundAnchorA.y -= y; jn.s1 -= x; jn.s2 -= y; }
	}
	for (size_t f = 0; f < w->fixtures.size(); ++f)
	{
		w->fixtures[f].fat[0] -= x; w->fixtures[f].fat[1] -= y;
		w->fixtures[f].fat[2] -= x; w->fixtures[f].fat[3] -= y;
	}
	rc = downloadState(w, 0);
	if (rc) return rc;
	w->stateCount = w->bodies.size();
	++w->mirrorEpoch;
	return B2HIP_OK;
}

int b2hip_joint_set_spring(b2hip_world* w, int joint, float frequency_hz, float damping_ratio)
{
	if (int rcu = checkUsable(w, "b2hip_joint_set_spring", true)) return rcu;
	if (joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	JointRec& j = w->joints[joint];
	if (j.type != B2D_JOINT_WHEEL && j.type != B2D_JOINT_DISTANCE && j.type != B2D_JOINT_WELD && j.type != B2D_JOINT_MOUSE)
		return setError(B2HIP_ERR_INVALID, "joint type has no spring");
	j.frequencyHz = frequency_hz;
	j.dampingRatio = damping_ratio;
	w->jointEdits.push_back(std::make_pair(joint, 0));
	return B2HIP_OK;
}

