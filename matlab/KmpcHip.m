classdef KmpcHip < Kmpc
    %KmpcHip: the reference's Kmpc with the per-step work on an MI355X (one kernel launch per MPC step).
    %   Drop-in: replace `Kmpc(` by `KmpcHip(` in example_control.m / Kmpc_setup.m (sysid_class must be a KsysidHip);
    %   Ksim.run_trial_mpc calls get_mpcInput / get_mpcInput_bilinear_iter exactly as before (Ksim.m:206-216) and still
    %   tests any(isnan(U)) for solver failure (:220).  Overridden: the three per-step entry points (Kmpc.m:329-387,
    %   750-814, 817-904: lift, get_costB/H/G/D_bilinear, constraint rows, quadprog).  The constructor of the parent
    %   still assembles its own cost / constraint matrices (used by nothing below, kept for inspection).
    %   mpc_type 'nonlinear' (fmincon) stays the parent's.

    properties
        hip;    % struct: mpc (uint64 handle), sys (the KsysidHip)
    end

    methods
        function obj = KmpcHip( sysid_class , varargin )
            obj = obj@Kmpc( sysid_class , varargin{:} );
            if ~isa( sysid_class , 'KsysidHip' )
                error( 'KmpcHip needs a KsysidHip (it shares its device context and dictionary)' );
            end
            obj.hip.sys = sysid_class;
            if strcmp( obj.mpc_type , 'nonlinear' )
                return;
            end
            m = obj.params.m;
            r = obj.cost_input(:);                       % eye(m).*cost_input (Kmpc.m:201,548): a column vector scales rows
            if numel( r ) == 1
                r = r * ones( m , 1 );
            end
            lo = []; hi = [];
            if ~isempty( obj.input_bounds )              % scaled-down bounds (:247,659)
                sc = obj.scaledown.u( obj.input_bounds' )';
                lo = sc(:,1); hi = sc(:,2);
            end
            slope = [];
            if ~isempty( obj.input_slopeConst )
                slope = obj.input_slopeConst * mean( obj.params.scale.u_factor );            % :272,684
            end
            smooth = [];
            if ~isempty( obj.input_smoothConst )
                smooth = obj.params.Ts^2 * obj.input_smoothConst * mean( obj.params.scale.u_factor );   % :294,706
            end
            mt = double( strcmp( obj.model_type , 'bilinear' ) );
            obj.hip.mpc = kp_mex( 'mpc_create' , sysid_class.hip.ctx , mt , obj.model.A , obj.model.B , obj.horizon , ...
                                  obj.projmtx , obj.cost_running , obj.cost_terminal , r , lo , hi , slope , smooth );
            if ~isempty( obj.state_bounds )              % :313
                sb = obj.scaledown.y( obj.state_bounds' )';
                kp_mex( 'mpc_set_state_bounds' , obj.hip.mpc , sb(:,1) , sb(:,2) );
            end
        end

        function [ U , z ] = hip_step( obj , traj , ref , iters )
            [ ~ , zeta_all ] = obj.get_zeta( traj );     % Kmpc.m:343-344
            zeta = zeta_all( end , : )';
            Np = obj.horizon;                            % reference padding (:354-365)
            if size( ref , 2 ) ~= size( obj.projmtx , 1 )
                error( 'Reference trajectory is not the correct dimension' );
            elseif size( ref , 1 ) > Np + 1
                ref = ref( 1 : Np + 1 , : );
            elseif size( ref , 1 ) < Np + 1
                ref = [ ref ; kron( ones( Np + 1 - size(ref,1) , 1 ) , ref(end,:) ) ];
            end
            Yr = reshape( ref' , [] , 1 );
            [ U , z ] = kp_mex( 'mpc_step_zeta' , obj.hip.mpc , obj.hip.sys.hip.basis , zeta , traj.u(end,:)' , Yr , iters );
        end

        function [ U , z ] = get_mpcInput( obj , traj , ref )
            [ U , z ] = obj.hip_step( traj , ref , 1 );                % Kmpc.m:329-387
        end

        function [ U , z ] = get_mpcInput_bilinear( obj , traj , ref )
            [ U , z ] = obj.hip_step( traj , ref , 1 );                % Kmpc.m:750-814
        end

        function [ U , z ] = get_mpcInput_bilinear_iter( obj , traj , ref , iter )
            [ U , z ] = obj.hip_step( traj , ref , iter );             % Kmpc.m:817-904
        end
    end
end
