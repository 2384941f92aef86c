// b2hip_api_snapshot.h - part of the ONE translation unit b2hip.hip, inside its extern "C" block: b2hip_save_snapshot /
// b2hip_load_snapshot (the binary checkpoint the reference lacks) and b2hip_get_contacts.
// (No include guard on purpose: b2hip.hip includes it exactly once, in order - the fragments share one scope.)

// ---- snapshot -----------------------------------------------------------------------------------------
namespace
{
struct SnapHeader
{
	char magic[8];
	uint32_t version, szDState, szHostBody, szHostFixture, szShape, szJoint;
	uint32_t nBodies, nFixtures, nShapes, nJoints, nFree, nContacts, nToiOrder, nMoves;
	uint32_t stateCount, cur;
	int32_t nextNode, leafCount, lastContacts, newFixture;
	float inv_dt0, cellSize;
	int32_t eventsOn, solverHints; // solverHints: bit 0 serialOrphansNext, bit 1 blocksTooBig, bit 2 step incomplete (sub-stepping), bits 3..7 + 31 freshColors, bits 8..15 adoptSticky, 16..23 largeHintSteps, 24..30 recolorCountdown (what the next island build is told)
};
const uint32_t kSnapVersion = 5;
const char kSnapMagic[8] = { 'B', '2', 'H', 'I', 'P', 'S', 'N', '1' };

struct SnapWriter
{
	std::vector<char> out;
	void host(const void* p, size_t n) { const char* c = (const char*)p; out.insert(out.end(), c, c + n); }
	int dev(const void* p, size_t n)
	{
		const size_t at = out.size();
		out.resize(at + n);
		if (n == 0) return 0;
		HIP_TRY(hipMemcpy(out.data() + at, p, n, hipMemcpyDeviceToHost));
		return 0;
	}
};

struct SnapReader
{
	const char* p;
	size_t left;
	bool ok;
	const void* take(size_t n)
	{
		if (n > left) { ok = false; return nullptr; }
		const void* r = p;
		p += n;
		left -= n;
		return r;
	}
	void host(void* dst, size_t n) { const void* s = take(n); if (s && n) memcpy(dst, s, n); }
	int dev(void* dst, size_t n)
	{
		const void* s = take(n);
		if (!s) return setError(B2HIP_ERR_INVALID, "snapshot truncated");
		if (n) HIP_TRY(hipMemcpy(dst, s, n, hipMemcpyHostToDevice));
		return 0;
	}
};
} // namespace

int b2hip_save_snapshot(b2hip_world* w, void* buffer, size_t cap, size_t* needed)
{
	if (!w || !needed || (cap > 0 && !buffer)) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rcu = checkUsable(w, "b2hip_save_snapshot", true)) return rcu;
	DEVICE_GUARD(w);
	ensureRows(w); // (the snapshot carries the host's mirror)
	int rc = flushEdits(w); // everything the host has created or edited is on the device now
	if (rc) return rc;
	rc = applyPendingFilters(w); // ... including the re-filter flags of joints created / destroyed since the last step
	if (rc) return rc;
	rc = applyEditOps(w, true);
	if (rc) return rc;
	rc = readState(w);
	if (rc) return rc;
	const DState& ds = *w->h_dstate;
	const size_t nb = w->bodies.size(), np = w->fixtures.size();
	const size_t nC = (size_t)std::max(ds.c.nContacts, 0), nM = (size_t)std::max(std::min(ds.c.nMoves, w->dw.capMoves), 0);
	const size_t nT = (size_t)std::max(ds.c.nToiOrder, 0);
	SnapHeader h;
	memset(&h, 0, sizeof(h));
	memcpy(h.magic, kSnapMagic, 8);
	h.version = kSnapVersion;
	h.szDState = sizeof(DState); h.szHostBody = (uint32_t)offsetof(HostBody, fixtures); h.szHostFixture = sizeof(HostFixture);
	h.szShape = sizeof(ShapeRec); h.szJoint = sizeof(RevoluteJoint);
	h.nBodies = (uint32_t)nb; h.nFixtures = (uint32_t)np; h.nShapes = (uint32_t)w->shapes.size(); h.nJoints = (uint32_t)w->joints.size();
	h.nFree = (uint32_t)w->freeUnits.size(); h.nContacts = (uint32_t)nC; h.nToiOrder = (uint32_t)nT; h.nMoves = (uint32_t)nM;
	h.stateCount = (uint32_t)std::min(w->stateCount, nb); h.cur = (uint32_t)ds.cur;
	h.nextNode = w->nextNode; h.leafCount = w->leafCount; h.lastContacts = w->lastContacts; h.newFixture = w->newFixture ? 1 : 0;
	h.inv_dt0 = w->inv_dt0; h.cellSize = w->dw.cellSize;
	h.eventsOn = w->eventsOn ? 1 : 0;
	h.solverHints = (w->serialOrphansNext ? 1 : 0) | (w->blocksTooBig ? 2 : 0) | (w->stepComplete ? 0 : 4) | ((w->adoptSticky & 0xff) << 8) | ((w->largeHintSteps & 0xff) << 16) | ((w->recolorCountdown & 0x7f) << 24) |
		((w->freshColors & 0x1f) << 3) | (int32_t)(((uint32_t)(w->freshColors >> 5) & 1u) << 31); // (bits 3..7 and 31: freshColors, 0..63)
	SnapWriter o;
	o.host(&h, sizeof(h));
	o.host(&w->def, sizeof(w->def));
	for (size_t i = 0; i < nb; ++i)
	{
		// (forces left in a row from before the last step are stale in an auto-clear world: the loaded world treats every
		// saved force as pending)
		HostBody row;
		memcpy((void*)&row, &w->bodies[i], offsetof(HostBody, fixtures));
		const HostBody& src = w->bodies[i];
		const bool current = src.dirty || src.pullEpoch == w->mirrorEpoch || i >= w->stateCount;
		if (w->def.auto_clear_forces && !(current && src.forceEpoch == w->stepEpoch) && i < w->stateCount) { row.fx = row.fy = row.torque = 0.0f; }
		o.host(&row, offsetof(HostBody, fixtures));
		const char dirty = w->bodies[i].dirty ? 1 : 0;
		o.host(&dirty, 1);
	}
	o.host(w->fixtures.data(), np * sizeof(HostFixture));
	o.host(w->shapes.data(), w->shapes.size() * sizeof(ShapeRec));
	o.host(w->freeUnits.data(), w->freeUnits.size() * sizeof(FreeUnit));
	o.host(w->h_state, (size_t)h.stateCount * 10 * sizeof(float));
	o.host(&ds, sizeof(DState));
#define SNAP_DEV(arr, n) do { rc = o.dev(w->arr.p, (size_t)(n) * sizeof(*w->arr.p)); if (rc) return rc; } while (0)
	SNAP_DEV(d_joints, w->joints.size()); // the device copy carries the accumulated impulses
	SNAP_DEV(b_pos, nb); SNAP_DEV(b_pos0, nb); SNAP_DEV(b_vel, nb); SNAP_DEV(b_xf, nb); SNAP_DEV(b_mass, nb); SNAP_DEV(b_damp, nb);
	SNAP_DEV(b_force, nb); SNAP_DEV(b_flags, nb); SNAP_DEV(b_wake, nb); SNAP_DEV(b_proxyHead, nb); SNAP_DEV(b_blk1, nb);
	SNAP_DEV(p_fat, np); SNAP_DEV(p_body, np); SNAP_DEV(p_shape, np); SNAP_DEV(p_key, np); SNAP_DEV(p_filter0, np); SNAP_DEV(p_filter1, np);
	SNAP_DEV(p_mat, np); SNAP_DEV(p_next, np);
	const int cur = ds.cur;
	SNAP_DEV(c_ids[cur], nC); SNAP_DEV(c_key[cur], nC); SNAP_DEV(c_flags[cur], nC); SNAP_DEV(c_mat[cur], nC); SNAP_DEV(c_man0[cur], nC);
	SNAP_DEV(c_man1[cur], nC); SNAP_DEV(c_imp[cur], nC); SNAP_DEV(c_man3[cur], nC); SNAP_DEV(c_color[cur], nC); SNAP_DEV(c_mgr[cur], nC);
	SNAP_DEV(toiPos2c, nT); SNAP_DEV(moveBuf, nM);
	{
		// trailing section (absent in snapshots of worlds saved before gear joints existed): the gear records
		const uint32_t tail[2] = { (uint32_t)w->gears.size(), (uint32_t)sizeof(GearRec) };
		o.host(tail, sizeof(tail));
		SNAP_DEV(d_gears, w->gears.size());
	}
#undef SNAP_DEV
	*needed = o.out.size();
	if (cap >= o.out.size()) memcpy(buffer, o.out.data(), o.out.size());
	else if (cap > 0) return setError(B2HIP_ERR_CAPACITY, "snapshot buffer too small");
	return B2HIP_OK;
}

// Everything in the blob is checked BEFORE anything is copied to the device or used as an index: the counts against each
// other and against the blob's size, every body / fixture / shape / joint / gear / proxy / contact index against its range.
// A truncated or bit-flipped snapshot is refused with B2HIP_ERR_INVALID; it never writes out of bounds.
int b2hip_load_snapshot(const void* buffer, size_t size, int device, b2hip_world** out)
{
	if (!buffer || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	SnapReader in = { (const char*)buffer, size, true };
	SnapHeader h;
	in.host(&h, sizeof(h));
	if (!in.ok || memcmp(h.magic, kSnapMagic, 8) != 0 || h.version != kSnapVersion || h.szDState != sizeof(DState) ||
		h.szHostBody != offsetof(HostBody, fixtures) || h.szHostFixture != sizeof(HostFixture) || h.szShape != sizeof(ShapeRec) ||
		h.szJoint != sizeof(RevoluteJoint))
		return setError(B2HIP_ERR_INVALID, "not a snapshot of this build of libb2hip");
	b2hip_world_def def;
	in.host(&def, sizeof(def));
	if (!in.ok) return setError(B2HIP_ERR_INVALID, "snapshot truncated");
	auto corrupt = [](const char* what) { return setError(B2HIP_ERR_INVALID, std::string("snapshot corrupt: ") + what); };
	const size_t nb = h.nBodies, np = h.nFixtures, nC = h.nContacts, nS = h.nShapes, nJ = h.nJoints, nT = h.nToiOrder, nM = h.nMoves;
	// no count can exceed what the blob could hold at all (this also keeps the size products below from overflowing)
	if (nb > size || np > size || nC > size || nS > size || nJ > size || h.nFree > size || nT > size || nM > size || h.stateCount > size)
		return corrupt("counts exceed the blob");
	if (h.stateCount > nb || h.cur > 1u || nT > nC || h.lastContacts < 0 || (size_t)h.lastContacts > nC) return corrupt("header counts");
	if (h.nextNode < 0 || h.leafCount < 0 || (size_t)h.leafCount > np || (size_t)h.nextNode > 2 * np + 2) return corrupt("proxy id allocator");
	if (!(h.cellSize > 0.0f) || !std::isfinite(h.cellSize)) return corrupt("cell size");

	// ---- host sections -------------------------------------------------------------------------------------------------
	const size_t bodyBytes = offsetof(HostBody, fixtures) + 1;
	const char* bodiesAt = (const char*)in.take(nb * bodyBytes);
	const HostFixture* fixturesAt = (const HostFixture*)in.take(np * sizeof(HostFixture));
	const ShapeRec* shapesAt = (const ShapeRec*)in.take(nS * sizeof(ShapeRec));
	const FreeUnit* freeAt = (const FreeUnit*)in.take((size_t)h.nFree * sizeof(FreeUnit));
	const float* stateAt = (const float*)in.take((size_t)h.stateCount * 10 * sizeof(float));
	const DState* dsAt = (const DState*)in.take(sizeof(DState));
	const RevoluteJoint* jointsAt = (const RevoluteJoint*)in.take(nJ * sizeof(RevoluteJoint));
	if (!in.ok) return setError(B2HIP_ERR_INVALID, "snapshot truncated");
	for (size_t i = 0; i < nb; ++i)
	{
		HostBody hb;
		memcpy((void*)&hb, bodiesAt + i * bodyBytes, offsetof(HostBody, fixtures));
		if (hb.type < 0 || hb.type > 2) return corrupt("body type");
	}
	for (size_t f = 0; f < np; ++f)
	{
		HostFixture hf;
		memcpy(&hf, fixturesAt + f, sizeof(hf));
		if (hf.body < 0 || (size_t)hf.body >= nb || hf.shape < 0 || (size_t)hf.shape >= nS || hf.proxyKey < 0) return corrupt("fixture");
	}
	for (size_t k = 0; k < nS; ++k)
	{
		ShapeRec sr;
		memcpy(&sr, shapesAt + k, sizeof(sr));
		if (sr.type < 0 || sr.type > 2 || sr.count < 0 || sr.count > B2D_MAX_POLY_VERTS) return corrupt("shape");
	}
	for (size_t k = 0; k < h.nFree; ++k)
	{
		FreeUnit fu;
		memcpy(&fu, freeAt + k, sizeof(fu));
		if (fu.leaf < -1 || fu.leaf >= h.nextNode) return corrupt("proxy id free list");
	}
	DState ds;
	memcpy(&ds, dsAt, sizeof(ds));
	if (ds.cur != (int)h.cur || ds.c.nContacts != (int)nC || ds.c.nToiOrder != (int)nT || ds.c.nMoves < (int)nM) return corrupt("device state block");

	// ---- device sections: located and range-checked in the blob, uploaded later ---------------------------------------------
	struct Sec { const void* p; size_t bytes; };
	auto sec = [&](size_t n, size_t elem) { Sec x = { in.take(n * elem), n * elem }; return x; };
	const Sec sPos = sec(nb, 16), sPos0 = sec(nb, 16), sVel = sec(nb, 16), sXf = sec(nb, 16), sMass = sec(nb, 16), sDamp = sec(nb, 16), sForce = sec(nb, 16);
	const Sec sFlags = sec(nb, 4), sWake = sec(nb, 4), sHead = sec(nb, 4), sBlk = sec(nb, 4);
	const Sec sFat = sec(np, 16), sPBody = sec(np, 4), sPShape = sec(np, 4), sPKey = sec(np, 4), sF0 = sec(np, 4), sF1 = sec(np, 4), sPMat = sec(np, 8), sNext = sec(np, 4);
	const Sec cIds = sec(nC, 16), cKey = sec(nC, 8), cFlags = sec(nC, 4), cMat = sec(nC, 16), cMan0 = sec(nC, 16), cMan1 = sec(nC, 16), cImp = sec(nC, 16),
		cMan3 = sec(nC, 16), cColor = sec(nC, 4), cMgr = sec(nC, 4);
	const Sec sToi = sec(nT, 4), sMoves = sec(nM, 4);
	if (!in.ok) return setError(B2HIP_ERR_INVALID, "snapshot truncated");
	auto inRange = [](const Sec& x, long long lo, long long hi) // every int of the section in [lo, hi)
	{
		const int* v = (const int*)x.p;
		for (size_t i = 0; i < x.bytes / 4; ++i)
		{
			int q;
			memcpy(&q, v + i, 4);
			if (q < lo || q >= hi) return false;
		}
		return true;
	};
	if (!inRange(sHead, -1, (long long)np) || !inRange(sNext, -1, (long long)np)) return corrupt("per-body proxy lists");
	if (!inRange(sBlk, 0, MAX_BLOCKS + 1) || ds.c.nBlocks < 0 || ds.c.nBlocks > MAX_BLOCKS) return corrupt("block partition");
	// (the proxy of a destroyed fixture stays in the table with body -1)
	if (!inRange(sPBody, -1, (long long)nb) || !inRange(sPShape, 0, (long long)nS)) return corrupt("proxy table");
	if (!inRange(sToi, 0, (long long)nC) || !inRange(sMoves, 0, (long long)np)) return corrupt("TOI order / move buffer");
	if (!inRange(cColor, -1, MAX_COLORS) || !inRange(cMgr, -1, (long long)std::max<size_t>(nT, 1))) return corrupt("contact colour / TOI slot");
	for (size_t i = 0; i < nC; ++i)
	{
		int4 ids;
		memcpy(&ids, (const char*)cIds.p + 16 * i, 16);
		if (ids.x < 0 || (size_t)ids.x >= np || ids.y < 0 || (size_t)ids.y >= np || ids.z < 0 || (size_t)ids.z >= nb || ids.w < 0 || (size_t)ids.w >= nb)
			return corrupt("contact ids");
	}
	// trailing section: the gear records (absent in snapshots of worlds that never had one)
	size_t nG = 0;
	const GearRec* gearsAt = nullptr;
	if (in.left >= 2 * sizeof(uint32_t))
	{
		uint32_t tail[2];
		in.host(tail, sizeof(tail));
		if (tail[1] != sizeof(GearRec) || tail[0] > size) return corrupt("gear section");
		nG = tail[0];
		gearsAt = (const GearRec*)in.take(nG * sizeof(GearRec));
		if (!gearsAt && nG) return setError(B2HIP_ERR_INVALID, "snapshot truncated");
	}
	for (size_t k = 0; k < nJ; ++k)
	{
		RevoluteJoint j;
		memcpy((void*)&j, jointsAt + k, sizeof(j));
		if (j.type < B2D_JOINT_DEAD || j.type > B2D_JOINT_GEAR) return corrupt("joint type");
		if (j.bodyA < 0 || (size_t)j.bodyA >= nb || j.bodyB < 0 || (size_t)j.bodyB >= nb) return corrupt("joint bodies");
		if (j.type == B2D_JOINT_GEAR && (j.enableLimit < 0 || (size_t)j.enableLimit >= nG)) return corrupt("gear index");
	}
	for (size_t k = 0; k < nG; ++k)
	{
		GearRec g;
		memcpy((void*)&g, gearsAt + k, sizeof(g));
		if (g.bodyC < 0 || (size_t)g.bodyC >= nb || g.bodyD < 0 || (size_t)g.bodyD >= nb) return corrupt("gear bodies");
	}

	// ---- build the world ---------------------------------------------------------------------------------------------------------
	def.device = device;
	b2hip_world* w = nullptr;
	int rc = b2hip_world_create(&def, &w);
	if (rc) return rc;
	DEVICE_GUARD(w);
	auto fail = [&](int code) { const std::string why = g_lastError; b2hip_world_destroy(w); return setError(code, why); };
	w->bodies.resize(nb);
	for (size_t i = 0; i < nb; ++i)
	{
		memcpy((void*)&w->bodies[i], bodiesAt + i * bodyBytes, offsetof(HostBody, fixtures));
		w->bodies[i].dirty = bodiesAt[i * bodyBytes + offsetof(HostBody, fixtures)] != 0;
		w->bodies[i].pullEpoch = 0;
		w->bodies[i].forceEpoch = w->stepEpoch;
		if (w->bodies[i].dirty) w->dirtyList.push_back((int)i);
	}
	{
		// m_nonStaticBodies from the saved slots
		size_t count = 0;
		for (size_t i = 0; i < nb; ++i) count += w->bodies[i].worldIndex >= 0 ? 1 : 0;
		w->nonStatic.assign(count, -1);
		for (size_t i = 0; i < nb; ++i)
		{
			const int k = w->bodies[i].worldIndex;
			if (k < 0) continue;
			if ((size_t)k >= count || w->nonStatic[(size_t)k] != -1) return fail(corrupt("non-static body order"));
			w->nonStatic[(size_t)k] = (int)i;
		}
		w->orderDirty = true;
	}
	w->fixtures.resize(np);
	if (np) memcpy(w->fixtures.data(), fixturesAt, np * sizeof(HostFixture));
	w->shapes.resize(nS);
	if (nS) memcpy((void*)w->shapes.data(), shapesAt, nS * sizeof(ShapeRec));
	w->freeUnits.resize(h.nFree);
	if (h.nFree) memcpy(w->freeUnits.data(), freeAt, (size_t)h.nFree * sizeof(FreeUnit));
	for (size_t f = 0; f < np; ++f) w->bodies[w->fixtures[f].body].fixtures.push_back((int)f);
	for (size_t k = 0; k < nS; ++k) w->shapeIndex[std::string((const char*)&w->shapes[k], sizeof(ShapeRec))] = (int)k;
	// joints / gears: the device copy is the truth (accumulated impulses); the host vectors mirror it and are uploaded, with
	// the per-body joint lists, by the next flushEdits
	w->joints.resize(nJ);
	if (nJ) memcpy((void*)w->joints.data(), jointsAt, nJ * sizeof(RevoluteJoint));
	for (size_t k = 0; k < nJ; ++k) w->nMouseJoints += w->joints[k].type == B2D_JOINT_MOUSE;
	w->gears.resize(nG);
	if (nG) memcpy((void*)w->gears.data(), gearsAt, nG * sizeof(GearRec));
	w->nextNode = h.nextNode; w->leafCount = h.leafCount; w->lastContacts = h.lastContacts; w->newFixture = h.newFixture != 0;
	w->inv_dt0 = h.inv_dt0;
	w->eventsOn = h.eventsOn != 0;
	w->serialOrphansNext = h.solverHints & 1;
	w->blocksTooBig = (h.solverHints & 2) != 0;
	w->stepComplete = (h.solverHints & 4) == 0;
	w->adoptSticky = (h.solverHints >> 8) & 0xff;
	w->largeHintSteps = (h.solverHints >> 16) & 0xff;
	w->recolorCountdown = (h.solverHints >> 24) & 0x7f;
	w->freshColors = ((h.solverHints >> 3) & 0x1f) | ((int)(((uint32_t)h.solverHints >> 31) & 1u) << 5);
	w->freshColorsPending = false;
	w->adoptPasses = w->adoptSticky > 0;
	rc = ensureCapacity(w, nC);
	if (rc) return fail(rc);
	if (nM > w->moveBuf.cap || nC > (size_t)w->dw.capContacts || (size_t)h.stateCount * 10 > w->h_stateCap) return fail(corrupt("counts exceed the buffers sized for them"));
	if (h.stateCount) memcpy(w->h_state, stateAt, (size_t)h.stateCount * 10 * sizeof(float));
	w->shadowDev = nullptr; // (the host's rows are the snapshot's now, not what the device last sent)
	w->stateCount = h.stateCount;
	*w->h_dstate = ds;
#define SNAP_UP(arr, s) do { if ((s).bytes && hipMemcpy(w->arr.p, (s).p, (s).bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(setError(B2HIP_ERR_HIP, "snapshot upload failed (" #arr ")")); } while (0)
	if (hipMemcpy(w->d_state.p, &ds, sizeof(DState), hipMemcpyHostToDevice) != hipSuccess) return fail(setError(B2HIP_ERR_HIP, "snapshot upload failed (state block)"));
	w->pubSeq = ds.pubCount; // (the device numbers its census publications; the host counts along)
	SNAP_UP(b_pos, sPos); SNAP_UP(b_pos0, sPos0); SNAP_UP(b_vel, sVel); SNAP_UP(b_xf, sXf); SNAP_UP(b_mass, sMass); SNAP_UP(b_damp, sDamp);
	w->forceOnDevice = true; w->rowsWentEarly = false;
	SNAP_UP(b_force, sForce); SNAP_UP(b_flags, sFlags); SNAP_UP(b_wake, sWake); SNAP_UP(b_proxyHead, sHead); SNAP_UP(b_blk1, sBlk);
	SNAP_UP(p_fat, sFat); SNAP_UP(p_body, sPBody); SNAP_UP(p_shape, sPShape); SNAP_UP(p_key, sPKey); SNAP_UP(p_filter0, sF0); SNAP_UP(p_filter1, sF1);
	SNAP_UP(p_mat, sPMat); SNAP_UP(p_next, sNext);
	const int cur = (int)h.cur;
	SNAP_UP(c_ids[cur], cIds); SNAP_UP(c_key[cur], cKey); SNAP_UP(c_flags[cur], cFlags); SNAP_UP(c_mat[cur], cMat); SNAP_UP(c_man0[cur], cMan0);
	SNAP_UP(c_man1[cur], cMan1); SNAP_UP(c_imp[cur], cImp); SNAP_UP(c_man3[cur], cMan3); SNAP_UP(c_color[cur], cColor); SNAP_UP(c_mgr[cur], cMgr);
	SNAP_UP(toiPos2c, sToi); SNAP_UP(moveBuf, sMoves);
#undef SNAP_UP
	if (nS && hipMemcpy(w->d_shapes.p, w->shapes.data(), nS * sizeof(ShapeRec), hipMemcpyHostToDevice) != hipSuccess)
		return fail(setError(B2HIP_ERR_HIP, "snapshot upload failed (shapes)"));
	w->upBodies = nb; w->upFixtures = np; w->upShapes = nS; w->upJoints = 0;
	w->dw.cellSize = h.cellSize;
	w->dw.invCellSize = 1.0f / h.cellSize;
	w->dw.eventsOn = w->eventsOn ? 1 : 0;
	*out = w;
	return B2HIP_OK;
}

int b2hip_get_contacts(b2hip_world* w, int cap, b2hip_contact* out)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	DEVICE_GUARD(w);
	if (!w || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	int rc = flushForRead(w);
	if (rc) return rc;
	rc = readState(w);
	if (rc) return rc;
	const int n = std::min(cap, w->h_dstate->c.nContacts);
	const int cur = w->h_dstate->cur;
	if (n <= 0) return 0;
	std::vector<int4> ids(n), m3(n);
	std::vector<uint32_t> flags(n);
	std::vector<float4> mat(n), m0(n), m1(n), imp(n);
	HIP_TRY(hipMemcpy(ids.data(), w->c_ids[cur].p, n * sizeof(int4), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(m3.data(), w->c_man3[cur].p, n * sizeof(int4), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(flags.data(), w->c_flags[cur].p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(mat.data(), w->c_mat[cur].p, n * sizeof(float4), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(m0.data(), w->c_man0[cur].p, n * sizeof(float4), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(m1.data(), w->c_man1[cur].p, n * sizeof(float4), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(imp.data(), w->c_imp[cur].p, n * sizeof(float4), hipMemcpyDeviceToHost));
	for (int i = 0; i < n; ++i)
	{
		b2hip_contact& c = out[i];
		c.fixture_a = ids[i].x;
		c.fixture_b = ids[i].y;
		c.body_a = ids[i].z;
		c.body_b = ids[i].w;
		c.flags = ((flags[i] & CF_TOUCHING) ? 1u : 0u) | ((flags[i] & CF_ENABLED) ? 2u : 0u);
		c.manifold_type = m3[i].z;
		c.point_count = m3[i].w;
		c.local_normal[0] = m0[i].x; c.local_normal[1] = m0[i].y;
		c.local_point[0] = m0[i].z; c.local_point[1] = m0[i].w;
		c.point_local[0][0] = m1[i].x; c.point_local[0][1] = m1[i].y;
		c.point_local[1][0] = m1[i].z; c.point_local[1][1] = m1[i].w;
		c.normal_impulse[0] = imp[i].x; c.tangent_impulse[0] = imp[i].y;
		c.normal_impulse[1] = imp[i].z; c.tangent_impulse[1] = imp[i].w;
		c.id_key[0] = (uint32_t)m3[i].x;
		c.id_key[1] = (uint32_t)m3[i].y;
		c.friction = mat[i].x;
		c.restitution = mat[i].y;
		c.tangent_speed = mat[i].z;
	}
	return n;
}

