// index.ts — HipBackend: the MI355X twin of CpuBackend behind tstwo's Backend / Column / PolyOps / FriOps / MerkleOps
// surface.  Intended location: packages/core/src/backend/hip/index.ts (imports below are relative to that place).
// NOT TESTED in the build image (no Bun); it is a line-for-line transcription of the Python mirror in tstwo_amd/
// (backend.py, poly.py, fri.py, vcs.py), which IS tested against the CPU oracle on an MI355X.
import { M31 } from "../../fields/m31";
import { QM31 } from "../../fields/qm31";
import type { Backend, Column } from "../index";
import { CirclePoint, Coset } from "../../circle";
import type { CircleDomain } from "../../poly/circle/domain";
import { CirclePoly } from "../../poly/circle/poly";
import { CircleEvaluation, type BitReversedOrder } from "../../poly/circle/evaluation";
import { TwiddleTree } from "../../poly/twiddles";
import { LineDomain, LineEvaluation } from "../../poly/line";
import { SecureEvaluation } from "../../poly/circle/secure_poly";
import { bitReverseIndex } from "../../utils";
import { DeviceBuffer, check, ensureInit, hip, ptr, ptrs, u32s } from "./ffi";

const q4 = (q: QM31): Uint32Array => Uint32Array.from(q.to_m31_array().map((m) => m.value));

/** Column<M31> resident in HBM as little-endian u32 words (M31.intoSlice layout). */
export class HipColumn implements Column<M31> {
  constructor(readonly buf: DeviceBuffer, private readonly n: number) {}
  static fromArray(data: readonly M31[]): HipColumn {           // value semantics: copies (cpu/index.ts:89)
    const b = new DeviceBuffer(4 * data.length);
    b.upload(Uint32Array.from(data, (m) => m.value));
    return new HipColumn(b, data.length);
  }
  static zeros(len: number): HipColumn {
    const b = new DeviceBuffer(4 * len);
    if (len) check(hip.tstwo_zero(b.dev, BigInt(4 * len)));
    return new HipColumn(b, len);
  }
  static uninitialized(len: number): HipColumn { return new HipColumn(new DeviceBuffer(4 * len), len); }
  get dev(): bigint { return this.buf.dev; }
  clone(): HipColumn {
    const out = HipColumn.uninitialized(this.n);
    if (this.n) check(hip.tstwo_copy(out.dev, this.dev, BigInt(4 * this.n)));
    return out;
  }
  len(): number { return this.n; }
  isEmpty(): boolean { return this.n === 0; }
  toU32(): Uint32Array { return this.buf.downloadU32(this.n); }
  toCpu(): M31[] { return Array.from(this.toU32(), (v) => M31.from_u32_unchecked(v)); }
  private checkIndex(i: number): void {
    if (!Number.isInteger(i) || i < 0 || i >= this.n) throw new Error(`Index ${i} out of bounds for column of length ${this.n}`);
  }
  at(i: number): M31 { this.checkIndex(i); return M31.from_u32_unchecked(this.buf.downloadU32(1, 4 * i)[0]!); }
  set(i: number, v: M31): void { this.checkIndex(i); this.buf.upload(Uint32Array.of(v.value), 4 * i); }
}

/** SecureColumnByCoords on the device: 4 coordinate columns (fields/secure_columns.ts:124). */
export class HipSecureColumn {
  constructor(readonly columns: [HipColumn, HipColumn, HipColumn, HipColumn]) {}
  static from(values: readonly QM31[]): HipSecureColumn {
    const cols = [0, 1, 2, 3].map((k) => HipColumn.fromArray(values.map((q) => q.to_m31_array()[k]!)));
    return new HipSecureColumn(cols as [HipColumn, HipColumn, HipColumn, HipColumn]);
  }
  static zeros(n: number): HipSecureColumn { return new HipSecureColumn([0, 1, 2, 3].map(() => HipColumn.zeros(n)) as any); }
  static uninitialized(n: number): HipSecureColumn { return new HipSecureColumn([0, 1, 2, 3].map(() => HipColumn.uninitialized(n)) as any); }
  len(): number { return this.columns[0].len(); }
  ptrs(): BigUint64Array { return ptrs(this.columns.map((c) => c.dev)); }
  at(i: number): QM31 { return QM31.from_m31_array(this.columns.map((c) => c.at(i)) as [M31, M31, M31, M31]); }
  to_vec(): QM31[] {
    const c = this.columns.map((x) => x.toU32());
    return Array.from({ length: this.len() }, (_, i) => QM31.from_u32_unchecked(c[0]![i]!, c[1]![i]!, c[2]![i]!, c[3]![i]!));
  }
}

export class HipBackend implements Backend {
  readonly name = "HipBackend";
  constructor() { ensureInit(); }
  bitReverseColumn(col: HipColumn): void {                           // throws "length is not power of two"
    check(hip.tstwo_bit_reverse(ptr(ptrs([col.dev])), 1n, BigInt(col.len())));
  }
  createBaseFieldColumn(data: M31[]): HipColumn { return HipColumn.fromArray(data); }
  createSecureFieldColumn(data: QM31[]): HipSecureColumn { return HipSecureColumn.from(data); }
  batchInverse(col: HipColumn): HipColumn {                          // throws "0 has no inverse"
    const out = HipColumn.uninitialized(col.len());
    check(hip.tstwo_m31_batch_inverse(col.dev, out.dev, BigInt(col.len())));
    return out;
  }
}

/** TwiddleTree with device buffers; generated on the GPU (backend/cpu/circle.ts:210-239). */
export function precomputeTwiddles(coset: Coset): TwiddleTree<HipBackend, HipColumn> {
  const n = coset.size();
  const tw = HipColumn.uninitialized(n), itw = HipColumn.uninitialized(n);
  check(hip.tstwo_twiddles_build(coset.initial_index.value, coset.log_size, tw.dev, itw.dev));
  return new TwiddleTree(coset, tw, itw);
}

function checkTree(domain: CircleDomain, tw: TwiddleTree<HipBackend, HipColumn>): void {
  if (!domain.halfCoset.is_doubling_of(tw.rootCoset)) throw new Error("twiddle tree mismatch");
}

// @ts-expect-error static-method dispatch contract of the reference (test/poly/circleEvaluation.test.ts:5-16)
export class HipCircleEvaluation extends CircleEvaluation<HipBackend, M31, BitReversedOrder> {
  constructor(domain: CircleDomain, readonly dev: HipColumn) { super(domain, dev as any); }
  static precomputeTwiddles = precomputeTwiddles;
  static to_cpu(values: HipColumn): M31[] { return values.toCpu(); }
  static bitReverseColumn(col: HipColumn): void { new HipBackend().bitReverseColumn(col); }
}

export class HipCirclePoly extends CirclePoly<HipBackend> {
  /** true reproduces the reference's log_size == 3 output swap (backend/cpu/circle.ts:123-131); default = Rust-exact. */
  static compatLog3Swap = false;
  constructor(readonly dev: HipColumn) { super(dev as any); }
  static precomputeTwiddles = precomputeTwiddles;

  static extend(poly: HipCirclePoly, logSize: number): HipCirclePoly {
    if (logSize < poly.logSize()) throw new Error("log size too small");
    const out = HipColumn.uninitialized(1 << logSize);
    check(hip.tstwo_poly_extend(poly.dev.dev, poly.logSize(), out.dev, logSize));
    return new HipCirclePoly(out);
  }
  static evaluate(poly: HipCirclePoly, domain: CircleDomain, tw: TwiddleTree<HipBackend, HipColumn>): HipCircleEvaluation {
    return HipCirclePoly.evaluatePolynomials([poly], domain, tw)[0]!;
  }
  /** Value semantics (the evaluation survives) with the copy folded into the first pass (tstwo_cfft_interpolate_to). */
  static interpolate(ev: HipCircleEvaluation, tw: TwiddleTree<HipBackend, HipColumn>): HipCirclePoly {
    checkTree(ev.domain, tw);
    const n = ev.domain.log_size();
    const col = HipColumn.uninitialized(1 << n);
    check(hip.tstwo_cfft_interpolate_to(ptr(ptrs([ev.dev.dev])), ptr(ptrs([col.dev])), 1n, n, ev.domain.halfCoset.initial_index.value, tw.itwiddles.dev, tw.rootCoset.log_size));
    return new HipCirclePoly(col);
  }
  /** PolyOps.evaluatePolynomials, batched (poly/circle/ops.ts:89-101): extend + evaluate per group of equal-sized polynomials
   *  without materialising the zero padding (tstwo_cfft_evaluate_extended). */
  static evaluatePolynomials(polys: HipCirclePoly[], domain: CircleDomain, tw: TwiddleTree<HipBackend, HipColumn>): HipCircleEvaluation[] {
    checkTree(domain, tw);
    const n = domain.log_size();
    const outs = polys.map((p) => {
      if (n < p.logSize()) throw new Error("log size too small");
      return HipColumn.uninitialized(1 << n);
    });
    const byLog = new Map<number, number[]>();
    polys.forEach((p, i) => byLog.set(p.logSize(), [...(byLog.get(p.logSize()) ?? []), i]));
    for (const [lg, idxs] of byLog) {
      check(hip.tstwo_cfft_evaluate_extended(ptr(ptrs(idxs.map((i) => polys[i]!.dev.dev))), lg, ptr(ptrs(idxs.map((i) => outs[i]!.dev))),
        BigInt(idxs.length), n, domain.halfCoset.initial_index.value, tw.twiddles.dev, tw.rootCoset.log_size));
    }
    return outs.map((c) => new HipCircleEvaluation(domain, c));
  }
  /** All polynomials of one size at one point in one launch sequence (prove_values' out-of-domain sampling). */
  static evalAtPointBatch(polys: HipCirclePoly[], point: CirclePoint<QM31>): QM31[] {
    const out = new Uint32Array(4 * polys.length);
    check(hip.tstwo_eval_at_point_batch(ptr(ptrs(polys.map((p) => p.dev.dev))), BigInt(polys.length), polys[0]!.logSize(), ptr(q4(point.x)), ptr(q4(point.y)), ptr(out)));
    return polys.map((_, i) => QM31.from_u32_unchecked(out[4 * i]!, out[4 * i + 1]!, out[4 * i + 2]!, out[4 * i + 3]!));
  }
  static eval_at_point(poly: HipCirclePoly, point: CirclePoint<QM31>): QM31 {
    const out = new Uint32Array(4);
    check(hip.tstwo_eval_at_point(poly.dev.dev, poly.logSize(), ptr(q4(point.x)), ptr(q4(point.y)), ptr(out)));
    return QM31.from_u32_unchecked(out[0]!, out[1]!, out[2]!, out[3]!);
  }
}

/** FriOps (fri.ts:93-110) with the reference's error texts. */
export class HipFriOps {
  fold_line(ev: { values: HipSecureColumn; domain(): LineDomain; len(): number }, alpha: QM31, tw?: TwiddleTree<HipBackend, HipColumn>) {
    const n = ev.len();
    if (n < 2) throw new Error("fold_line: Evaluation too small, must have at least 2 elements.");
    const domain = ev.domain(), k = domain.logSize();
    const out = HipSecureColumn.uninitialized(n / 2);
    if (tw && domain.coset().is_doubling_of(tw.rootCoset)) {
      check(hip.tstwo_fri_fold_line(ptr(ev.values.ptrs()), k, tw.itwiddles.dev, tw.rootCoset.log_size, ptr(q4(alpha)), ptr(out.ptrs())));
    } else {   // domain not covered by a precomputed tree: the n/2 inverses the reference computes per element
      const inv = HipColumn.fromArray(Array.from({ length: n / 2 }, (_, i) => domain.at(bitReverseIndex(i << 1, k)).inverse()));
      check(hip.tstwo_fri_fold_line_tw(ptr(ev.values.ptrs()), k, inv.dev, ptr(q4(alpha)), ptr(out.ptrs())));
    }
    return LineEvaluation.new(domain.double(), out as any);
  }
  fold_circle_into_line(dst: { values: HipSecureColumn; len(): number }, src: { values: HipSecureColumn; domain: CircleDomain }, alpha: QM31, tw?: TwiddleTree<HipBackend, HipColumn>): void {
    if ((src.domain.size() >> 1) !== dst.len()) throw new Error("fold_circle_into_line: Length mismatch between src and dst after considering fold step.");
    const n = src.domain.log_size();
    if (tw && n >= 3 && src.domain.halfCoset.is_doubling_of(tw.rootCoset)) {
      check(hip.tstwo_fri_fold_circle_into_line(ptr(dst.values.ptrs()), BigInt(dst.len()), ptr(src.values.ptrs()), n, tw.itwiddles.dev, tw.rootCoset.log_size, ptr(q4(alpha))));
    } else {
      const inv = HipColumn.fromArray(Array.from({ length: dst.len() }, (_, i) => src.domain.at(bitReverseIndex(i << 1, n)).y.inverse()));
      check(hip.tstwo_fri_fold_circle_into_line_tw(ptr(dst.values.ptrs()), BigInt(dst.len()), ptr(src.values.ptrs()), n, inv.dev, ptr(q4(alpha))));
    }
  }
  decompose(ev: { values: HipSecureColumn; domain: CircleDomain }): [SecureEvaluation<HipBackend, BitReversedOrder>, QM31] {
    const n = ev.values.len();
    const out = HipSecureColumn.uninitialized(n), lam = new Uint32Array(4);
    check(hip.tstwo_fri_decompose(ptr(ev.values.ptrs()), BigInt(n), ptr(out.ptrs()), ptr(lam)));
    return [new SecureEvaluation(ev.domain, out as any), QM31.from_u32_unchecked(lam[0]!, lam[1]!, lam[2]!, lam[3]!)];
  }
}

/** MerkleOps<Blake2sHash>: layers stay in HBM; hashNode semantics (children AND the layer's column values). */
export class HipMerkleOps {
  commitOnLayer(logSize: number, prevLayer: DeviceBuffer | undefined, columns: readonly HipColumn[]): DeviceBuffer {
    const out = new DeviceBuffer(32 << logSize);
    check(hip.tstwo_merkle_commit_layer(logSize, prevLayer ? prevLayer.dev : 0n, ptr(ptrs(columns.map((c) => c.dev))), BigInt(columns.length), out.dev));
    return out;
  }
  /** MerkleProver.commit: every layer, root first (layer k at byte 32*(2^k-1)), plus the 32-byte root on the host. */
  commit(columns: readonly HipColumn[]): { layers: DeviceBuffer; root: Uint8Array } {
    const logs = columns.map((c) => Math.log2(c.len()));
    const maxLog = columns.length ? Math.max(...logs) : 0;
    const layers = new DeviceBuffer(32 * ((2 << maxLog) - 1)), root = new Uint8Array(32);
    check(hip.tstwo_merkle_commit(ptr(ptrs(columns.map((c) => c.dev))), ptr(u32s(logs)), BigInt(columns.length), layers.dev, ptr(root)));
    return { layers, root };
  }
  /** MerkleProver.decommit (vcs/prover.ts:32-109) on a tree built by commit(): the walk and both gathers run in the library. */
  decommit(layers: DeviceBuffer, columns: readonly HipColumn[], queriesPerLogSize: Map<number, number[]>):
      { queriedValues: M31[]; hashWitness: Uint8Array[]; columnWitness: M31[] } {
    const logs = columns.map((c) => Math.log2(c.len()));
    const maxLog = columns.length ? Math.max(...logs) : 0;
    const sets = [...queriesPerLogSize].filter(([, q]) => q.length > 0);
    const totalQ = sets.reduce((a, [, q]) => a + q.length, 0);
    const capV = Math.max(1, totalQ * Math.max(1, columns.length)), capH = Math.max(1, 2 * totalQ * (maxLog + 1));
    const qArrays = sets.map(([, q]) => BigUint64Array.from(q.map(BigInt)));
    const qPtrs = BigUint64Array.from(qArrays.map((a) => BigInt(ptr(a))));
    const nQ = BigUint64Array.from(sets.map(([, q]) => BigInt(q.length)));
    const queried = new Uint32Array(capV), colWit = new Uint32Array(capV), hashes = new Uint8Array(32 * capH);
    const counts = BigUint64Array.from([BigInt(capV), BigInt(capH), BigInt(capV)]);
    check(hip.tstwo_merkle_decommit(layers.dev, maxLog, ptr(ptrs(columns.map((c) => c.dev))), ptr(u32s(logs)), BigInt(columns.length),
      ptr(u32s(sets.map(([lg]) => lg))), ptr(qPtrs), ptr(nQ), BigInt(sets.length),
      ptr(queried), ptr(counts.subarray(0, 1)), ptr(hashes), ptr(counts.subarray(1, 2)), ptr(colWit), ptr(counts.subarray(2, 3))));
    const [nq, nh, nw] = [Number(counts[0]), Number(counts[1]), Number(counts[2])];
    return {
      queriedValues: Array.from(queried.subarray(0, nq), (v) => M31.from_u32_unchecked(v)),
      hashWitness: Array.from({ length: nh }, (_, i) => hashes.slice(32 * i, 32 * i + 32)),
      columnWitness: Array.from(colWit.subarray(0, nw), (v) => M31.from_u32_unchecked(v)),
    };
  }
}
