//! `BabyBearPoseidon2HipEngine`: the type `crates/prover` aliases behind the `hip` feature
//! (`crates/prover/src/prover/mod.rs:27-39`: `DeferralEngine`, `VerifyProver`, `VerifyCircuitProver`; constructed by
//! `<E as StarkEngine>::new(SystemParams)` at mod.rs:209).
//!
//! SKETCH, not compiled: the trait surface below is recalled from the pinned `openvm-stark-backend` v2.0.0, whose
//! source is not vendored in the reference tree; the bodies state what each method hands to the C ABI.
//! Two gaps remain before this can replace the CUDA engine for real: the pinned backend's
//! protocol is the v2 sum-check/WHIR stack while libzkhip implements the v1 quotient + FRI pipeline the task names,
//! and `Proof<SC>`'s byte layout (`Encode`) is not this library's.
use crate::{AirDesc, HipContext, Params, ProvingKey};

pub struct BabyBearPoseidon2HipEngine {
    pub ctx: HipContext,
    pub params: Params,
}

impl BabyBearPoseidon2HipEngine {
    /// `StarkEngine::new(params)` (mod.rs:209): one context on the current device.
    pub fn new(params: Params) -> Self {
        // upstream panics on a missing device (CudaError); keep that contract
        let ctx = HipContext::new(0).expect("no gfx950 device for the HIP engine");
        Self { ctx, params }
    }

    /// `StarkEngine::keygen` -> device proving key: every chip's symbolic constraints are lowered once to the
    /// bytecode of `AirDesc::program` (walk of `SymbolicConstraintsDag` nodes: Variable{Main|Preprocessed, offset} ->
    /// VAR/PREP, IsFirstRow/IsLastRow/IsTransition, Add/Sub/Mul/Neg, Constant, PublicValue; interactions ->
    /// the trailing section), then `zkhip_keygen`.
    pub fn keygen<'c>(&'c self, airs: &[AirDesc]) -> crate::Result<ProvingKey<'c>> {
        ProvingKey::keygen(&self.ctx, &self.params, airs)
    }

    /// `StarkEngine::prove(pk, ctx)`: traces are the device matrices the chips' trace generators produced
    /// (column-major Montgomery u32 -- the layout both backends use); public values go through `as_canonical_u32()`.
    pub fn prove(&self, pk: &ProvingKey<'_>, traces: &[&crate::DeviceBuffer<'_>], pvs: &[Vec<u32>]) -> crate::Result<Vec<u8>> {
        pk.prove(traces, pvs)
    }
}
